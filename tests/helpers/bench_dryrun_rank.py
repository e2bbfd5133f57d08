"""One rank of bench.py's N > 1 path on the CPU: the same `bench.run` (rank / world
from the launcher's environment, barrier-fenced timed loop, metric reduction,
rank-0 JSON line), with gloo instead of RCCL and a toy workload coded by the CPU
oracle instead of the HIP engine.  Started by tests/test_bench_launcher.py through
`bench.launch_ranks` -- test infrastructure, not a product path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch

import bench


class OracleToyWorkload(object):
    """F random symbol planes per step through EntEncoder -> file -> EntDecoder on the oracle"""

    name = "codec"

    def __init__(self, args, rank, local, dev):
        from pseudocylindrical_convolution_amd.PCONV_operator import backend
        from oracle import pconv_cpu, coder_cpu
        backend.use(pconv_cpu, coder_cpu)
        pconv_cpu.set_detmath(True)
        from pseudocylindrical_convolution_amd import pseudo_codec as PC
        torch.manual_seed(1234)
        self.enc = PC.EntEncoder(4, 16, True, 8, gid=0)
        self.dec = PC.EntDecoder(4, 16, True, 8, gid=0)
        g = torch.Generator().manual_seed(7)
        sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in self.enc.state_dict().items()}
        self.enc.load_state_dict(sd)
        self.dec.load_state_dict(sd)
        self.H, self.W, self.F = 256, 1024, args.frames_per_gpu   # symbol planes (16, 4, 1, 64)
        self.rank = rank
        self.syms = [torch.randint(0, 8, (16, 4, 1, 64), generator=torch.Generator().manual_seed(100 + rank * self.F + i)).float()
                     for i in range(self.F)]
        self.dir = os.environ["PCONV_DRYRUN_DIR"]
        import json
        # what bench.pin_rank left this rank, and how the native engine would size its host side here (the two
        # entry points touch no GPU: csrc/engine.cpp allowed_cpus / step_pool_spin_us)
        from pseudocylindrical_convolution_amd import _native
        lib = _native.hip_lib()
        with open(os.path.join(self.dir, "affinity_r%d.json" % rank), "w") as f:
            json.dump({"cpus": sorted(os.sched_getaffinity(0)), "frames": self.F, "torch_threads": torch.get_num_threads(),
                       "engine_host_cpus": lib.pconv_ee_host_cpus(), "engine_spin_us_8_frames": lib.pconv_ee_spin_us(8),
                       "local_world": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))}, f)
        self.bits, self.bits_first, self.back = 0, None, None

    def step(self):
        self.bits, self.back = 0, []
        for i, sym in enumerate(self.syms):
            path = os.path.join(self.dir, "r%d_f%d.bin" % (self.rank, i))
            self.enc.start(path)
            self.enc(sym.clone())
            self.dec.start(path)
            self.back.append(self.dec(1, 64))
            self.bits += os.path.getsize(path) * 8
        if self.bits_first is None:
            self.bits_first = self.bits

    def pixels_per_step(self):
        return float(self.F) * self.H * self.W

    def check(self):
        assert self.bits == self.bits_first
        for sym, back in zip(self.syms, self.back):
            assert torch.equal(back, self.enc.fill(sym.clone()))
        return {"psnr_sum": 0.0, "ssim_sum": 0.0}

    def describe(self):
        return "CPU dry run: %d toy symbol plane(s) per rank per step on the oracle" % self.F


if __name__ == "__main__":
    bench.run(bench.parse_args(), workload_cls=OracleToyWorkload, dist_backend="gloo", device_type="cpu")
