"""Timeline of 64 consecutive workgroups of a 1x1 / GDN launch (profiling build, -DPCONV_CONV_STAMP):
   PCONV_HIP_LIB=tools/_build/libpconv_hip_cstamp.so python tools/gpu_probe_conv_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pseudocylindrical_convolution_amd import PCONV, _native
from pseudocylindrical_convolution_amd.PCONV_operator import set_weight
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W16 = np.asarray(set_weight(16, True), dtype=np.float32)
lib = _native.hip_lib()


def report(title):
    out = (ctypes.c_ulonglong * (64 * 8 * 6))()
    assert lib.pconv_conv_read_stamps(out) == 0
    a = np.array(list(out), dtype=np.float64).reshape(64, 8, 6)
    t0 = a[:, 0, 0].min()
    print(title)
    print("  cycles from the first of the 64 workgroups' entry; wave 0 of each workgroup; cu = HW_ID (se, cu, simd ...)")
    print("  block   hw_id     entry  ->first chunk  ->loop end  ->done   | prologue  loop  way out")
    order = np.argsort(a[:, 0, 0])
    for i in order[:48]:
        e, f, l, d = a[i, 0, 0] - t0, a[i, 0, 1] - t0, a[i, 0, 2] - t0, a[i, 0, 3] - t0
        print("  %6d  %08x  %8.0f  %8.0f  %8.0f  %8.0f  | %6.0f  %6.0f  %6.0f" % (
            a[i, 0, 5], int(a[i, 0, 4]), e, f, l, d, f - e, l - f, d - l))
    pro, loop, way = a[:, 0, 1] - a[:, 0, 0], a[:, 0, 2] - a[:, 0, 1], a[:, 0, 3] - a[:, 0, 2]
    print("  mean cycles: prologue %.0f, matrix loop %.0f, way out %.0f" % (pro.mean(), loop.mean(), way.mean()))


for (cin, cout, rows, cols, res) in ((96, 192, 32, 1024, True),):
    x = torch.randn(16, cin, rows, cols, device=dev)
    conv = torch.nn.Conv2d(cin, cout, 1).to(dev)
    r = torch.randn(16, cout, rows, cols, device=dev) if res else None
    lim = torch.from_numpy(PCONV.tile_widths(W16, 16, rows * 16, cols).astype(np.int32)).to(dev)
    for _ in range(3):
        PCONV.tile_conv2d(conv, x, conv.weight, conv.bias, 1, None, lim, 16, residual=r, trim=res)
    torch.cuda.synchronize()
    report("1x1 %d->%d w%d%s" % (cin, cout, cols, " +residual" if res else ""))
for (ch, rows, cols) in ((192, 64, 2048),):
    x = torch.randn(16, ch, rows, cols, device=dev)
    gamma = (torch.rand(ch, ch, device=dev) * 0.01 + torch.eye(ch, device=dev) * 0.1).contiguous()
    beta = torch.ones(ch, device=dev)
    r = torch.randn(16, ch, rows, cols, device=dev)
    lim = torch.from_numpy(PCONV.tile_widths(W16, 16, rows * 16, cols).astype(np.int32)).to(dev)
    owner = torch.nn.Module()
    for _ in range(3):
        PCONV.tile_gdn(owner, x, gamma, beta, False, lim, 16, r, 0)
    torch.cuda.synchronize()
    report("GDN %d w%d +residual" % (ch, cols))
