"""GPU probe: the encoder's hidden layers on the matrix cores (csrc/entropy_mfma.hip) against the vector kernel
(PCONV_EE_BULK=valu): identical streams, entropy-encode seconds for N frames of random symbols.
  python tools/gpu_probe_entropy_mfma.py N [reps] [H W]      (H x W = rows per tile x columns of the symbol tensor)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudocylindrical_convolution_amd.engine import EntropyEngine
from pseudocylindrical_convolution_amd import pseudo_codec as PC
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16
W = int(sys.argv[4]) if len(sys.argv) > 4 else 512
torch.manual_seed(1234)
enc = PC.PseudoEncoder(56, 0)
g = torch.Generator().manual_seed(7)
sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
enc.ent.load_state_dict(sd)
sym = torch.randint(0, 8, (16 * N, 14, H, W), generator=torch.Generator().manual_seed(3)).float().cuda()
sym = enc.ent.fill(sym).contiguous()
res = {}
for mode in ("valu", "mfma", "mfma4"):
    os.environ["PCONV_EE_BULK"] = mode[:4]
    os.environ["PCONV_EE_MFMA_FORM"] = "4b" if mode == "mfma4" else "16x4"
    e = EntropyEngine(enc.ent, H, W, N, "cuda:0")
    best = 1e9
    for rep in range(reps):
        torch.cuda.synchronize(); t0 = time.time()
        streams = e.encode(sym); torch.cuda.synchronize(); t1 = time.time()
        best = min(best, t1 - t0)
    res[mode] = (streams, best)
    print("N=%d %dx%d %s: entropy encode %.4f s, bytes %d" % (N, H, W, mode, best, len(streams[0])), flush=True)
same = res["valu"][0] == res["mfma"][0] == res["mfma4"][0]
print("streams identical:", same, flush=True)
sys.exit(0 if same else 1)
