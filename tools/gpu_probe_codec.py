"""GPU probe: full codec on the HIP backend; parity vs the CPU oracle at 256x512,
then timing at larger sizes.  Writes progress lines (flush) for long runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudocylindrical_convolution_amd.PCONV_operator import backend
from pseudocylindrical_convolution_amd import pseudo_codec as PC

def build(vd=56):
    torch.manual_seed(1234)
    enc = PC.PseudoEncoder(vd, 0); dec = PC.PseudoDecoder(vd, 0)
    g = torch.Generator().manual_seed(7)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd); dec.ent.load_state_dict(sd)
    dec.quant.weight.data.copy_(enc.quant.weight.data)
    return enc, dec

def run(enc, dec, x, path, tag):
    dev = x.device
    sync = (lambda: torch.cuda.synchronize()) if x.is_cuda else (lambda: None)
    sync(); t0 = time.time(); sym = enc.symbols(x); sync(); t1 = time.time()
    enc.ent.start(path); enc.ent(sym.clone()); sync(); t2 = time.time()
    dec.ent.start(path); out = dec.ent(sym.shape[2], sym.shape[3]); sync(); t3 = time.time()
    rec = dec.reconstruct(out); sync(); t4 = time.time()
    ok = torch.equal(out, enc.ent.fill(sym.clone()))
    print("%s: analysis %.3fs entropy-enc %.3fs entropy-dec %.3fs synthesis %.3fs bytes %d roundtrip %s"
          % (tag, t1 - t0, t2 - t1, t3 - t2, t4 - t3, os.path.getsize(path), ok), flush=True)
    return sym, out, rec

sizes = [(256, 512), (512, 1024)] + ([(2048, 4096)] if "--big" in sys.argv else [])
enc, dec = build()
res = {}
for (H, W) in sizes:
    x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1))
    for rep in range(2):
        res[(H, W)] = run(enc, dec, x.cuda(), "/tmp/gpu_%d.bin" % H, "gpu %dx%d rep%d" % (H, W, rep))
if "--vendor" in sys.argv:
    os.environ["PCONV_TILE_CONV"] = "vendor"
    for (H, W) in sizes:
        x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1))
        for rep in range(2):
            run(enc, dec, x.cuda(), "/tmp/gpuv_%d.bin" % H, "vendor-conv %dx%d rep%d" % (H, W, rep))
    os.environ["PCONV_TILE_CONV"] = "native"
# parity against the oracle at the smallest size
from oracle import pconv_cpu, coder_cpu
backend.use(pconv_cpu, coder_cpu); pconv_cpu.set_detmath(True)
cenc, cdec = build()
H, W = sizes[0]
x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1))
csym, cout, crec = run(cenc, cdec, x, "/tmp/cpu_%d.bin" % H, "oracle %dx%d" % (H, W))
gsym, gout, grec = res[(H, W)]
print("symbols equal:", torch.equal(gsym.cpu(), csym), " mismatches:", (gsym.cpu() != csym).sum().item(), "of", csym.numel())
a, b = open("/tmp/gpu_%d.bin" % H, "rb").read(), open("/tmp/cpu_%d.bin" % H, "rb").read()
print("bitstreams equal:", a == b, len(a), len(b))
print("recon max abs diff:", (grec.cpu() - crec).abs().max().item())
