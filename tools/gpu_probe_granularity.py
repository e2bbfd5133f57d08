import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (rows, cols) in ((32, 1024), (512, 64), (256, 128), (8, 4096)):
    for (cin, cout, res) in ((96, 192, True), (192, 192, True)):
        x = torch.randn(16, cin, rows, cols, device=dev)
        conv = torch.nn.Conv2d(cin, cout, 1).to(dev)
        r = torch.randn(16, cout, rows, cols, device=dev)
        ms = timed(lambda: PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, None, None, 0, residual=r))
        px = 16.0 * rows * cols
        print("1x1 %d->%d %dx%d +residual: %.3f ms  %.2f TB/s" % (cin, cout, rows, cols, ms, px * 4 * (cin + 2 * cout) / ms * 1e-9))
