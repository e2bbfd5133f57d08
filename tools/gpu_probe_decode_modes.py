"""GPU probe: entropy decode of N frames, queued vs host-driven chain, with the engine's own timing split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudocylindrical_convolution_amd.engine import CodecEngine
from pseudocylindrical_convolution_amd import pseudo_codec as PC
torch.manual_seed(1234)
enc, dec = PC.PseudoEncoder(56, 0), PC.PseudoDecoder(56, 0)
g = torch.Generator().manual_seed(7)
sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
enc.ent.load_state_dict(sd); dec.ent.load_state_dict(sd); dec.quant.weight.data.copy_(enc.quant.weight.data)
eng = CodecEngine(56, 0, enc, dec)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.rand(N, 3, 2048, 4096, generator=torch.Generator().manual_seed(1)).cuda()
sym = eng.symbols(x)
e = eng._engine("enc", sym.shape[2], sym.shape[3], N)
streams = e.encode(sym.contiguous())
d = eng._engine("dec", sym.shape[2], sym.shape[3], N)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = d.decode(streams); torch.cuda.synchronize()
    print("N=%d decode %.1f ms ok %s" % (N, (time.time() - t0) * 1e3, torch.equal(out, sym)), flush=True)
