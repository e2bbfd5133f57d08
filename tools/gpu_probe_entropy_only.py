"""GPU probe: the native entropy engine alone (no transforms) on random symbols of a 2048x4096 frame:
  python tools/gpu_probe_entropy_only.py N [reps]
prints encode / decode seconds for N frames in one call; meant to run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudocylindrical_convolution_amd.engine import CodecEngine
from pseudocylindrical_convolution_amd import pseudo_codec as PC
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "both"   # both | encode
torch.manual_seed(1234)
enc, dec = PC.PseudoEncoder(56, 0), PC.PseudoDecoder(56, 0)
g = torch.Generator().manual_seed(7)
sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
enc.ent.load_state_dict(sd); dec.ent.load_state_dict(sd)
eng = CodecEngine(56, 0, enc, dec)
sym = torch.randint(0, 8, (16 * N, 14, 16, 512), generator=torch.Generator().manual_seed(3)).float().cuda()
sym = enc.ent.fill(sym).contiguous()
e = eng._engine("enc", 16, 512, N)
d = eng._engine("dec", 16, 512, N)
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.time()
    streams = e.encode(sym); torch.cuda.synchronize(); t1 = time.time()
    out = d.decode(streams) if mode == "both" else sym; torch.cuda.synchronize(); t2 = time.time()
    print("N=%d rep%d: entropy encode %.4f s, decode %.4f s, ok %s, bytes %d" % (N, rep, t1 - t0, t2 - t1, torch.equal(out, sym), len(streams[0])), flush=True)
