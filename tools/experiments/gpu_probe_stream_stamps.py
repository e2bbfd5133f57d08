"""Where a chunk period of the streamed 1x1 kernel goes (profiling build, -DPCONV_STREAM_STAMP):
   PCONV_HIP_LIB=tools/_build/libpconv_hip_sstamp.so PCONV_CONV1X1=stream python tools/gpu_probe_stream_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV, _native
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W16 = np.asarray(set_weight(16, True), dtype=np.float32)
lib = _native.hip_lib()
for (tn, cin, cout, rows, cols, res) in ((128, 96, 192, 32, 1024, True), (128, 192, 96, 34, 1026, False), (128, 192, 192, 32, 1024, True)):
    x = torch.randn(tn, cin, rows, cols, device=dev)
    conv = torch.nn.Conv2d(cin, cout, 1).to(dev)
    r = torch.randn(tn, cout, rows, cols, device=dev) if res else None
    lim = torch.from_numpy(PCONV.tile_widths(W16, 16, rows * 16, cols).astype(np.int32)).to(dev)
    for _ in range(2):
        PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, None, lim, 16, residual=r, trim=res)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * (16 * 64 * 4))()
    assert lib.pconv_stream_read_stamps(out) == 0
    a = np.array(list(out), dtype=np.float64).reshape(16, 64, 4)
    print("1x1 %d->%d %dx%d x%d%s: one workgroup, chunk periods 16..47 (matrix work of a SIMD's two waves: 3072 cycles per period)" % (
        cin, cout, rows, cols, tn, " +residual" if res else ""))
    P = slice(16, 48)
    per = np.diff(a[0, 16:49, 0]).mean()
    print("  period (wave 0, start to start): %.0f cycles" % per)
    print("  matrix wave   matrix block   stage wait   at barrier   (cycles per period, mean of 32)")
    for w in (0, 4, 1, 5, 3, 7):
        m = a[w, P]
        print("     %2d         %8.0f      %8.0f     %8.0f" % (w, (m[:, 1] - m[:, 0]).mean(), (m[:, 2] - m[:, 1]).mean(), (m[:, 3] - m[:, 2]).mean()))
    print("  drain wave    slice work     at barrier")
    for w in (8, 12, 9, 15):
        m = a[w, P]
        print("     %2d         %8.0f      %8.0f" % (w, (m[:, 1] - m[:, 0]).mean(), (m[:, 2] - m[:, 1]).mean()))
    t0 = a[0, 16, 0]
    print("  one period in detail (cycles since wave 0's period start): wave: start, block issued, wait over, barrier passed")
    for w in (0, 4, 8, 12):
        m = a[w, 20]
        print("     %2d: " % w + "  ".join("%7.0f" % (v - a[0, 20, 0]) for v in m[: (4 if w < 8 else 3)]))
