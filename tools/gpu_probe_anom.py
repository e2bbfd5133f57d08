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
H, W, N = 2048, 4096, 1
x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
def lap(tag, t0):
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ms = torch.cuda.memory_stats()
    print("  %-12s host %7.1f total %7.1f ms | dev allocs %d frees %d reserved %.1f GB" % (tag, (t1 - t0) * 1e3, (t2 - t0) * 1e3, ms["num_device_alloc"], ms["num_device_free"], ms["reserved_bytes.all.current"] / 2**30), flush=True)
    return time.perf_counter()
for rep in range(4):
    print("rep", rep)
    torch.cuda.synchronize(); t = time.perf_counter()
    sym = enc.symbols(x); t = lap("enc.symbols", t)
    sym = enc.ent.fill(sym); t = lap("fill", t)
    e = eng._engine("enc", sym.shape[2], sym.shape[3], N)
    streams = e.encode(sym.contiguous()); t = lap("ent-enc", t)
    d = eng._engine("dec", sym.shape[2], sym.shape[3], N)
    out = d.decode(streams); t = lap("ent-dec", t)
    rec = dec.reconstruct(out); t = lap("synthesis", t)
    del rec, out
