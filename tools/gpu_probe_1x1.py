"""Time the weight-resident 1x1 / GDN kernels at the codec's shapes (frames of 4096x2048)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W16 = np.asarray(set_weight(16, True), dtype=np.float32)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (cin, cout, rows, cols, res) in ((96, 192, 32, 1024, True), (192, 96, 34, 1026, False), (192, 192, 32, 1024, True), (192, 768, 32, 1024, False)):
    TN = int(os.environ.get("PROBE_TN", "16"))   # 128: the codec's batched small-scale layers (8 frames)
    x = torch.randn(TN, cin, rows, cols, device=dev)
    conv = torch.nn.Conv2d(cin, cout, 1).to(dev)
    r = torch.randn(TN, cout, rows, cols, device=dev) if res else None
    wd = PCONV.tile_widths(W16, 16, rows * 16, cols)
    lim = torch.from_numpy(wd.astype(np.int32)).to(dev)
    ms = timed(lambda: PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, None, lim, 16, residual=r, trim=res))
    px = float(wd.sum()) * rows * TN / 16
    byt = px * 4 * (cin + cout + (cout if res else 0))
    print("1x1 %d->%d w%d%s: %.3f ms  %.2f TB/s  %.1f TFLOP/s (valid columns)" % (
        cin, cout, cols, " +residual" if res else "", ms, byt / ms * 1e-9, 2.0 * cin * cout * px / ms * 1e-9))
for (ch, rows, cols) in ((192, 64, 2048), (192, 32, 1024)):
    x = torch.randn(16, ch, rows, cols, device=dev)
    gamma = (torch.rand(ch, ch, device=dev) * 0.01 + torch.eye(ch, device=dev) * 0.1).contiguous()
    beta = torch.ones(ch, device=dev)
    r = torch.randn(16, ch, rows, cols, device=dev)
    wd = PCONV.tile_widths(W16, 16, rows * 16, cols)
    lim = torch.from_numpy(wd.astype(np.int32)).to(dev)
    owner = torch.nn.Module()
    ms = timed(lambda: PCONV.tile_gdn(owner, x, gamma, beta, False, lim, 16, r, 0))
    px = float(wd.sum()) * rows
    print("GDN %d w%d +residual: %.3f ms  %.2f TB/s (x once) %.1f TFLOP/s" % (
        ch, cols, ms, px * 4 * 3 * ch / ms * 1e-9, 2.0 * ch * ch * px / ms * 1e-9))
