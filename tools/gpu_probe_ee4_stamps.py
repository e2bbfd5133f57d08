"""GPU probe: where a workgroup of the four-block encoder kernel spends its time (stamp build:
tools/build_variant.sh ee4stamp entropy_mfma.hip -DPCONV_EE4_STAMP; PCONV_HIP_LIB points at it): wall-clock stamps
(100 MHz) at entry / patch landed / loop done / stores issued of every workgroup of the LAST hidden-layer launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ["PCONV_EE_BULK"] = "mfma"
os.environ["PCONV_EE_MFMA_FORM"] = "4b"
os.environ["PCONV_ENGINE_ENCODE_RANGES"] = "1"
from pseudocylindrical_convolution_amd.engine import EntropyEngine
from pseudocylindrical_convolution_amd import pseudo_codec as PC
from pseudocylindrical_convolution_amd import _native
H, W = 16, 512
torch.manual_seed(1234)
enc = PC.PseudoEncoder(56, 0)
g = torch.Generator().manual_seed(7)
enc.ent.load_state_dict({k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()})
sym = torch.randint(0, 8, (16, 14, H, W), generator=torch.Generator().manual_seed(3)).float().cuda()
sym = enc.ent.fill(sym).contiguous()
e = EntropyEngine(enc.ent, H, W, 1, "cuda:0")
for rep in range(3):
    e.encode(sym)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["PCONV_HIP_LIB"])
n = 5 * 8192
buf = np.zeros(n, dtype=np.int64)
rc = lib.pconv_debug_ee4_stamps(buf.ctypes.data_as(ctypes.c_void_p), n)
assert rc == 0, rc
st = buf.reshape(8192, 5)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = (st[:, :4] - t0) / 100.0
print("workgroups stamped:", len(st), " launch span %.1f us" % us[:, 3].max())
pro, loop, out = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
for name, v in (("entry -> patch landed", pro), ("loop", loop), ("way out (stores issued)", out), ("whole", us[:, 3] - us[:, 0])):
    print("%-26s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, v.mean(), *np.percentile(v, [10, 50, 90]), v.max()))
hw = st[:, 4] & 0xffffffff
xcc = (st[:, 4] >> 32) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
unit = xcc * 1000 + se * 100 + sh * 16 + cu
units = np.unique(unit)
print("distinct CUs seen:", len(units))
# per CU: busy span and the number of workgroups
cnt = np.array([np.sum(unit == u) for u in units])
last = np.array([us[unit == u, 3].max() for u in units])
first = np.array([us[unit == u, 0].min() for u in units])
print("workgroups per CU: min %d mean %.1f max %d;  last exit per CU: min %.1f mean %.1f max %.1f us;  first entry: max %.1f us"
      % (cnt.min(), cnt.mean(), cnt.max(), last.min(), last.mean(), last.max(), first.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("XCC %d: %4d workgroups, last exit %.1f us" % (x, m.sum(), us[m, 3].max()))
# concurrency on one CU: time line of the busiest
u = units[np.argmax(cnt)]
sel = np.argsort(us[unit == u, 0])
print("one CU (%d workgroups): entry / landed / loop done / out" % cnt.max())
for row in us[unit == u][sel][:24]:
    print("   %7.1f %7.1f %7.1f %7.1f" % tuple(row))
