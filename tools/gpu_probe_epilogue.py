import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pseudocylindrical_convolution_amd import PCONV
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cin, cout, k, rows, cols, reps = 192, 192, 3, 64, 2048, 5
x = torch.randn(16, cin, rows + k - 1, cols + k - 1, device=dev)
conv = torch.nn.Conv2d(cin, cout, k, 1).to(dev)
slope = torch.full((cout,), 0.25, device=dev)
res = torch.randn(16, cout, rows, cols, device=dev)
for name, bias, sl, kw in (("bias+prelu", conv.bias, slope, {}), ("no bias, no act", None, None, {}), ("bias only", conv.bias, None, {}),
                           ("bias+prelu+residual", conv.bias, slope, {"residual": res})) * 2:
    for _ in range(2):
        y = PCONV.tile_conv2d(conv, x, conv.weight, bias, 1, sl, None, 0, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = PCONV.tile_conv2d(conv, x, conv.weight, bias, 1, sl, None, 0, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-22s %.3f ms  %.1f TFLOP/s" % (name, ms, 2.0 * cin * 9 * cout * y.numel() / cout / ms * 1e-9))
