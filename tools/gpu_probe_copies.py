"""GPU probe: which Python lines of a bench step launch library copies / fills (torch profiler with stacks).
  python tools/gpu_probe_copies.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse_args(["--steps", "1", "--warmup", "0"])
args.height, args.width = args.height or 2048, args.width or 4096
args.frames_per_gpu = args.frames_per_gpu or 8
torch.cuda.set_device(0)
load = bench.CodecWorkload(args, 0, 0, "cuda:0")
load.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    load.step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::zeros", "aten::clone", "aten::contiguous", "aten::cat"):
        dev = getattr(ev, "device_time_total", None)
        if dev is None:
            dev = getattr(ev, "cuda_time_total", 0)
        stack = [s for s in (ev.stack or []) if "pseudocylindrical" in s or "bench.py" in s][:3]
        rows.append((dev, ev.name, tuple(stack)))
agg = {}
for dev, name, stack in rows:
    k = (name, stack)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += dev
for (name, stack), (n, dev) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-18s x%-4d %9.1f us device  %s" % (name, n, dev, " <- ".join(s.strip()[-90:] for s in stack)))
