"""Phase cycles of one workgroup of the Winograd F(4x2,3x3) kernel (profiling build, -DPCONV_W42_STAMP):
   PCONV_HIP_LIB=tools/_build/libpconv_hip_w42stamp.so python tools/gpu_probe_wino42_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV, _native
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W16 = np.asarray(set_weight(16, True), dtype=np.float32)
os.environ["PCONV_CONV3X3"] = "wino42!"
lib = _native.hip_lib()
for (tn, cin, cout, rows, cols, res, d2w) in ((16, 192, 192, 64, 2048, True, False), (16, 192, 768, 32, 1024, False, True)):
    x = torch.randn(tn, cin, rows + 2, cols + 2, device=dev)
    conv = torch.nn.Conv2d(cin, cout, 3).to(dev)
    sl = torch.rand(cout, device=dev)
    r = torch.randn(tn, cout, rows, cols, device=dev) if res else None
    lim = torch.from_numpy(PCONV.tile_widths(W16, 16, rows * 16, cols).astype(np.int32)).to(dev)
    for _ in range(3):
        PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, sl, lim, 16, residual=r, trim=res, d2w=d2w, ring=2)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 48)()
    assert lib.pconv_wino42_read_stamps(out) == 0
    a = np.array(list(out), dtype=np.float64).reshape(8, 6)
    nch = cin // 4
    print("3x3 %d->%d %dx%d%s: cycles of one workgroup (matrix work of a SIMD's two waves: %d per chunk, %d chunks = %d)" % (
        cin, cout, rows, cols, " d2w" if d2w else "", 3072, nch, 3072 * nch))
    print("  wave  prologue  main loop  (per chunk)  way out   total    barrier wait per steady chunk")
    for w in range(8):
        print("  %d    %8.0f  %9.0f  %11.0f  %7.0f  %7.0f    %6.0f" % (w, a[w, 1] - a[w, 0], a[w, 2] - a[w, 1], (a[w, 2] - a[w, 1]) / nch,
              a[w, 3] - a[w, 2], a[w, 3] - a[w, 0], a[w, 4] / max(a[w, 5], 1)))
