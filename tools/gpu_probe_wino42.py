"""Time Winograd F(4x2,3x3) (csrc/wino42.hip) against F(2x2,3x3) (csrc/wino.hip) and check both against the direct
fp32-MFMA kernel at the codec's 3x3 stride-1 shapes (one 4096x2048 frame; the 1/4-scale layers also batched)."""
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


cases = [  # tn, cin, cout, rows (out), cols (out), residual+trim, d2w
    (16, 192, 192, 64, 2048, True, False), (16, 192, 192, 64, 2048, False, False), (16, 192, 192, 32, 1024, True, False),
    (128, 192, 192, 32, 1024, True, False), (16, 96, 96, 32, 1024, False, False), (128, 96, 96, 32, 1024, False, False),
    (16, 192, 768, 32, 1024, False, True), (128, 192, 768, 16, 512, False, True), (128, 192, 192, 16, 512, True, False),
]
if os.environ.get("PCONV_PROBE_SHORT"):
    cases = [cases[0], cases[4], cases[6]]
modes = ("direct", "wino", "wino42!") if not os.environ.get("PCONV_PROBE_NODIRECT") else ("wino", "wino42!")
for (tn, cin, cout, rows, cols, res, d2w) in cases:
    x = torch.randn(tn, cin, rows + 2, cols + 2, device=dev)
    conv = torch.nn.Conv2d(cin, cout, 3).to(dev)
    sl = torch.rand(cout, device=dev)
    r = torch.randn(tn, cout, rows, cols, device=dev) if res else None
    wd = PCONV.tile_widths(W16, 16, rows * 16, cols)
    lim = torch.from_numpy(wd.astype(np.int32)).to(dev)
    px = float(wd.sum()) * rows * (tn // 16)
    out = {}
    for mode in modes:
        os.environ["PCONV_CONV3X3"] = mode
        ms = timed(lambda: PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, sl, lim, 16, residual=r, trim=res, d2w=d2w, ring=2))
        out[mode] = (ms, PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, sl, lim, 16, residual=r, trim=res, d2w=d2w).clone())
    ref = out[modes[0]][1]
    fl = 2.0 * cin * 9 * cout * px
    print("3x3 %d->%d tn%d %dx%d%s%s: " % (cin, cout, tn, rows, cols, " +res" if res else "", " d2w" if d2w else "") +
          "  ".join("%s %.3f ms (%.0f TF alg, diff %.2g)" % (m, out[m][0], fl / out[m][0] * 1e-9, (out[m][1] - ref).abs().max().item())
                    for m in modes), flush=True)
