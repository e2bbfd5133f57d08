"""GPU probe for rocprofv3 --pmc passes over the decoder's step kernel: ONE lock-step group of two frames (the shape
of each of the benchmark's four groups), one encode + `reps` decodes at 4096x2048.  PCONV_EE_XCD=0/1 selects the
workgroup order of ee_step_kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PCONV_ENGINE_GROUPS", "1")
import torch
from pseudocylindrical_convolution_amd.engine import CodecEngine
from pseudocylindrical_convolution_amd import pseudo_codec as PC
torch.manual_seed(1234)
enc, dec = PC.PseudoEncoder(56, 0), PC.PseudoDecoder(56, 0)
g = torch.Generator().manual_seed(7)
sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
enc.ent.load_state_dict(sd); dec.ent.load_state_dict(sd)
eng = CodecEngine(56, 0, enc, dec)
N, reps = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 1
sym = torch.randint(0, 8, (16 * N, 14, 16, 512), generator=torch.Generator().manual_seed(3)).float().cuda()
sym = enc.ent.fill(sym).contiguous()
streams = eng._engine("enc", 16, 512, N).encode(sym)
d = eng._engine("dec", 16, 512, N)
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.time()
    out = d.decode(streams); torch.cuda.synchronize()
    print("one group of %d frames: decode %.1f ms ok %s (PCONV_EE_XCD=%s)" % (N, (time.time() - t0) * 1e3, torch.equal(out, sym),
                                                                          os.environ.get("PCONV_EE_XCD", "1")), flush=True)
