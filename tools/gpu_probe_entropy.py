"""GPU probe: native entropy engine only (random symbols), for profiling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudocylindrical_convolution_amd.engine import EntropyEngine
from pseudocylindrical_convolution_amd import pseudo_codec as PC
H, W, N = 2048, 4096, 1
for a in sys.argv[1:]:
    if a.startswith("--size="): H, W = (int(v) for v in a[7:].split("x"))
    if a.startswith("--n="): N = int(a[4:])
torch.manual_seed(1234)
ent = PC.EntEncoder(14, 16, True, 8, gid=0)
g = torch.Generator().manual_seed(7)
ent.load_state_dict({k: torch.randn(v.shape, generator=g) * 0.05 for k, v in ent.state_dict().items()})
h, w = PC.latent_shape(H, W)
sym = torch.randint(2, 6, (N * 16, 14, 2 * h, 2 * w), generator=g).float().cuda()
sym = ent.fill(sym)
eng = EntropyEngine(ent, 2 * h, 2 * w, N, "cuda:0")
for rep in range(2 if "--once" not in sys.argv else 1):
    torch.cuda.synchronize(); t0 = time.time()
    streams = eng.encode(sym); torch.cuda.synchronize(); t1 = time.time()
    out = eng.decode(streams) if "--enc-only" not in sys.argv else sym; torch.cuda.synchronize(); t2 = time.time()
    print("%dx%d N=%d: enc %.3f dec %.3f s bytes %s ok %s" % (H, W, N, t1 - t0, t2 - t1, [len(s) for s in streams], torch.equal(out, sym)), flush=True)
