import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pseudocylindrical_convolution_amd import pseudo_codec as PC
torch.manual_seed(1234)
enc = PC.PseudoEncoder(56, 0)
H, W = 2048, 4096
x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
def T(fn, name, res):
    torch.cuda.synchronize(); t0 = time.perf_counter(); y = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    res.append((name, (t1 - t0) * 1e3, (t2 - t0) * 1e3)); return y
with torch.no_grad():
    for rep in range(3):
        res = []
        ms0 = torch.cuda.memory_stats()
        y = T(lambda: enc.slice(x), "slice", res)
        for i, m in enumerate(enc.encoder.net):
            y = T(lambda: m(y), "net.%d %s" % (i, type(m).__name__), res)
        y = T(lambda: enc.encoder.trim(enc.encoder.act(y)), "sigmoid+trim", res)
        q = T(lambda: enc.quant(y), "quant", res)
        ms1 = torch.cuda.memory_stats()
        print("rep", rep, "device allocs %d frees %d retries %d" % (ms1["num_device_alloc"] - ms0["num_device_alloc"], ms1["num_device_free"] - ms0["num_device_free"], ms1["num_alloc_retries"] - ms0["num_alloc_retries"]))
        for n, h, t in res:
            print("   %-32s host %8.2f ms  total %8.2f ms" % (n, h, t))
        print("   sum total %.1f ms" % sum(t for _, _, t in res), flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); s = enc.symbols(x); torch.cuda.synchronize(); print("symbols() %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    torch.cuda.synchronize(); t0 = time.perf_counter(); s = enc.symbols(x); torch.cuda.synchronize(); print("symbols() %.1f ms" % ((time.perf_counter() - t0) * 1e3))
