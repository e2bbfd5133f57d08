"""Phase cycles of the Winograd kernel's steady-state chunk (profiling build, -DPCONV_WINO_STAMP):
   PCONV_HIP_LIB=tools/_build/libpconv_hip_stamp.so python tools/gpu_probe_wino_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV, _native
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W16 = np.asarray(set_weight(16, True), dtype=np.float32)
os.environ["PCONV_CONV3X3"] = "wino"
lib = _native.hip_lib()
for (tn, cin, cout, rows, cols, res) in ((16, 192, 192, 64, 2048, True), (16, 96, 96, 32, 1024, False)):
    x = torch.randn(tn, cin, rows + 2, cols + 2, device=dev)
    conv = torch.nn.Conv2d(cin, cout, 3).to(dev)
    sl = torch.rand(cout, device=dev)
    r = torch.randn(tn, cout, rows, cols, device=dev) if res else None
    lim = torch.from_numpy(PCONV.tile_widths(W16, 16, rows * 16, cols).astype(np.int32)).to(dev)
    for _ in range(3):
        PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, sl, lim, 16, residual=r, trim=res, ring=2)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 48)()
    assert lib.pconv_wino_read_stamps(out) == 0
    a = np.array(list(out), dtype=np.float64).reshape(8, 6)
    print("3x3 %d->%d %dx%d: cycles per steady chunk (matrix work of a SIMD's two waves: 3072)" % (cin, cout, rows, cols))
    print("  wave  barrier  patchDMA  matrix  weightDMA  -  total   (chunks)")
    for w in range(8):
        n = max(a[w, 5], 1)
        v = a[w, :5] / n
        print("  %d    %7.0f  %9.0f  %8.0f  %6.0f  %9.0f  %6.0f   (%d)" % (w, v[0], v[1], v[2], v[3], v[4], v.sum(), a[w, 5]))
