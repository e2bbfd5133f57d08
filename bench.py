#!/usr/bin/env python3
"""Headline benchmark: ERP MPix/s, encode + decode, 4096x2048 frames, model-idx 3
(--ssim => valid_dim 56), synthetic frames and seeded random weights.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no WORLD_SIZE in the environment this process starts the N
ranks itself (torch.distributed.run as a child, before anything here touches the
GPU) and exits with their code.  One step = every rank encodes and decodes its
own shard of frames (--frames-per-gpu, independent frames, no data-path
collective: weak scaling).  Rank 0 prints ONE JSON line.  Besides the contract
fields it carries
  roofline      the dominant kernel (the Winograd F(4x2,3x3) tile convolution on the
                fp32 matrix cores, csrc/wino42.hip) timed with events on its launch
                stream during the timed steps.  `achieved` / `frac` count the
                multiply-adds the algorithm EXECUTES on the matrix cores (24 per
                4x2 outputs, input and output channel), so frac <= 1 is a fraction
                of the fp32 MFMA peak; `direct_equivalent` is the same time priced
                in direct-convolution flops (x 3, SURVEY 8d's per-pixel figure).
                `traffic` / `mfma_busy` from the rocprofv3 --pmc summary under
                profiles/ when that summary is of the same kernel
  hbm           the gather / permute kernels of the same steps against the HBM roof
  cpu_baseline  the CPU oracle port of the same codec on a bounded sample

--mode analysis times the analysis transform alone (BASELINE config #3:
SphereSlice + EncoderV2, 1x3x1024x2048) and reports a roofline table per kernel
class instead.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

VALID_FRACTION = 836.0 / 1024.0      # valid columns / all columns
MFMA_F32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md, dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md, HBM3E
WINOGRAD_GAIN = {"wino_": 2.25,      # F(2x2,3x3): 16 matrix multiply-adds per 2x2 outputs instead of 36
                 "wino42_": 3.0}      # F(4x2,3x3): 24 per 4x2 outputs instead of 72


def winograd_gain(kernel):
    """direct-convolution flops per executed matrix-core flop of a tile-conv kernel (1 for the direct kernels)"""
    for prefix, gain in WINOGRAD_GAIN.items():
        if kernel.startswith(prefix):
            return gain
    return 1.0
MODEL_VALID_DIM = 56                 # model-idx 3 of the --ssim list (pseudo_codec.py:18-19)
PMC_SUMMARIES = [os.path.join(ROOT, "profiles", n) for n in ("round5_bench_pmc.json", "round4_bench_pmc.json", "round3_bench_pmc.json",
                                                             "round2_bench_pmc.json")]


# ----------------------------------------------------------------------------
# launcher
# ----------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_cpus(local_rank, local_world, allowed=None):
    """the slice of the allowed CPUs that rank `local_rank` of `local_world` ranks on this node
    keeps: contiguous, disjoint, every rank at least one (the rest of the division goes to the
    first ranks).  With fewer CPUs than ranks the ranks share all of them."""
    cpus = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    if local_world <= 1 or len(cpus) < local_world:
        return cpus
    base, extra = divmod(len(cpus), local_world)
    lo = local_rank * base + min(local_rank, extra)
    return cpus[lo:lo + base + (1 if local_rank < extra else 0)]


def cpu_quota():
    """CPUs the cgroup lets this job use (a GPU box shows all host CPUs in the affinity mask of a
    container that owns a share of them); None when there is no quota"""
    override = os.environ.get("PCONV_CGROUP_CPU_MAX")   # another cpu.max-format file (tests; csrc/engine.cpp reads it too)
    try:
        with open(override or "/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        if override:
            return None
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        return max(1, quota // period) if quota > 0 else None
    except (OSError, ValueError):
        return None


def pin_rank(local_rank, local_world, emulate=False):
    """Before this rank creates any thread or touches the GPU: keep it (and every thread it starts:
    torch's intra-op pool, the engine's queueing / polling / coder threads) on its own slice of the
    host cores, so that 8 ranks x (4 drivers + up to 8 coder threads + OpenMP) do not migrate over
    each other's cores.  PCONV_BENCH_PIN=0 turns it off.  Returns the number of cores kept.
    emulate (--emulate-local-world): this is the ONLY rank, but it gets what rank 0 of `local_world`
    ranks would get at best -- its affinity slice cut down to quota / local_world CPUs, so that the
    threads really compete for a rank's share of the host."""
    if not hasattr(os, "sched_setaffinity") or os.environ.get("PCONV_BENCH_PIN", "1") == "0":
        return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cpus = rank_cpus(local_rank, local_world)
    quota = cpu_quota()
    n = max(1, len(cpus) if quota is None else min(len(cpus), max(1, quota // max(local_world, 1))))
    if emulate:
        cpus = cpus[:n]
    os.sched_setaffinity(0, cpus)
    # ... and the threads that exist already (numpy's BLAS pool is started by `import torch`): a rank's share of
    # the host holds for ALL its threads
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
    except OSError:
        pass
    n = min(n, 32)   # the host side of a rank is a handful of driver / coder threads; torch's pool never needs more
    os.environ["OMP_NUM_THREADS"] = str(n)
    torch.set_num_threads(n)
    return n


def thread_table():
    """[(name, allowed CPUs, user + system seconds)] of every thread of this process (diagnostic of
    --emulate-local-world: which threads burn the rank's share of the host, and whether any escaped the pin)"""
    rows, tick = [], os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            base = "/proc/self/task/%s/" % tid
            with open(base + "stat") as f:
                st = f.read()
            name = st[st.index("(") + 1:st.rindex(")")]
            fields = st[st.rindex(")") + 2:].split()
            cpu = (int(fields[11]) + int(fields[12])) / tick
            allowed = ""
            with open(base + "status") as f:
                for line in f:
                    if line.startswith("Cpus_allowed_list"):
                        allowed = line.split(":")[1].strip()
            rows.append((name, allowed, cpu))
    except (OSError, ValueError, IndexError):
        pass
    return rows


def launch_ranks(nproc, argv, script=None, env=None):
    """Start `nproc` ranks of `script` (this file) under torch.distributed.run, one per
    GPU, and return their exit code.  The caller must not have initialised the GPU: the
    ranks are CHILD processes (never an exec of this one), each picks its device from
    LOCAL_RANK before its first HIP call."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           script or os.path.abspath(__file__)] + list(argv)
    child_env = dict(os.environ)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child_env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(nproc, 1))))
    if env:
        child_env.update(env)
    return subprocess.call(cmd, env=child_env)


# ----------------------------------------------------------------------------
# workload pieces
# ----------------------------------------------------------------------------
def make_codec(device_id, vd=MODEL_VALID_DIM):
    """seeded random-weight codec in a FIXED state: eval mode (the quantiser's training-mode
    level merge, pseudo_quant_cuda.cu:97-143, must not fire inside a benchmark) and the
    decoder's level table tied to the encoder's"""
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    torch.manual_seed(1234)
    enc, dec = PC.PseudoEncoder(vd, device_id).eval(), PC.PseudoDecoder(vd, device_id).eval()
    g = torch.Generator().manual_seed(7)
    # the reference's default torch.rand init makes degenerate CDFs
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd)
    dec.ent.load_state_dict(sd)
    with torch.no_grad():
        dec.quant.weight.copy_(enc.quant.weight)
    return enc, dec


def synthetic_frame(h, w, seed, device):
    """smooth low-frequency ERP frame plus a little noise, in [0, 1]"""
    g = torch.Generator().manual_seed(seed)
    yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
    ph = torch.rand(3, 4, generator=g) * 6.28318
    chans = []
    for c in range(3):
        v = 0.5 + 0.2 * torch.sin(6.28318 * (2 + c) * xx + ph[c, 0]) * torch.cos(3.14159 * (1 + c) * yy + ph[c, 1]) \
            + 0.15 * torch.sin(6.28318 * 7 * xx + 12.566 * yy + ph[c, 2])
        chans.append(v)
    img = torch.cat(chans, 1) + 0.04 * torch.rand(1, 3, h, w, generator=g)
    return img.clamp_(0, 1).to(device).contiguous()


class ConvProbe(object):
    """collects the tile-conv / GDN launches of the timed steps: PCONV appends
    (kernel, class label, algorithmic flops, start event, end event)"""

    def __init__(self):
        self.records = []

    def summarise_bytes(self):
        """per kernel of the HBM-bound ops: algorithmic bytes, seconds, launches, and the same per class"""
        return self.summarise()

    def summarise(self):
        torch.cuda.synchronize()
        per = {}
        for kernel, label, flops, e0, e1, *rest in self.records:
            d = per.setdefault(kernel, {"flops": 0.0, "seconds": 0.0, "launches": 0, "classes": {}})
            t = e0.elapsed_time(e1) * 1e-3
            d["flops"] += flops
            d["seconds"] += t
            d["launches"] += 1
            c = d["classes"].setdefault(label, [0.0, 0.0, 0, 0.0])
            c[0] += flops
            c[1] += t
            c[2] += 1
            c[3] += rest[0] if rest else 0.0   # algorithmic HBM bytes (tile-conv / GDN records)
        return per


def _pmc_record(kernel):
    for path in PMC_SUMMARIES:
        if not os.path.exists(path):
            continue
        with open(path) as f:
            pmc = json.load(f)
        rec = pmc.get("kernels", {}).get(kernel)
        if rec:
            return rec, path
    return None, None


def pmc_evidence(kernel, avg_flops):
    """HBM bytes per launch and matrix-pipe busy fraction of `kernel` from the tracked
    rocprofv3 --pmc summary (tools/summarise_pmc.py over separate counter passes of this
    very command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).  PMC counters
    cannot be read from inside the process; a summary of another kernel is not used."""
    rec, path = _pmc_record(kernel)
    if not rec:
        return {}
    out = {"traffic_source": "profiles/%s" % os.path.basename(path)}
    if rec.get("hbm_bytes_per_launch") is not None:
        out["traffic"] = int(rec["hbm_bytes_per_launch"])
        if rec.get("algorithmic_flops_per_launch"):
            # same kernel, possibly a different mix of launch sizes: scale by flops
            out["traffic"] = int(rec["hbm_bytes_per_launch"] * avg_flops / rec["algorithmic_flops_per_launch"])
    if rec.get("mfma_busy") is not None:
        out["mfma_busy"] = rec["mfma_busy"]
    return out


def hbm_table(per_kernel):
    """rows for the HBM-bound gather / permute kernels (north_star: slice / pseudo_pad / pseudo_fill /
    uslice ... against the HBM roof): algorithmic bytes of SURVEY 8d per launch / event time"""
    rows = []
    for kernel, d in sorted(per_kernel.items(), key=lambda kv: -kv[1]["seconds"]):
        n = d["launches"]
        gbs = d["flops"] / d["seconds"] / 1e9            # the probe's work column holds bytes here
        row = {"kernel": kernel, "launches": n, "avg_launch_us": round(d["seconds"] / n * 1e6, 2),
               "mb_per_launch": round(d["flops"] / n / 1e6, 3), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None}
        rec, path = _pmc_record(kernel)
        if rec and rec.get("hbm_bytes_per_launch") is not None:
            # counted bytes per launch of the --pmc pass, rescaled to THIS run's launch size by the algorithmic
            # bytes (the same kernel over another number of frames); a summary without the pass's algorithmic
            # bytes (rounds 2-4: taken at 2 frames per GPU) cannot be rescaled and is not quoted as this run's
            if rec.get("algorithmic_bytes_per_launch"):
                row["traffic"] = int(rec["hbm_bytes_per_launch"] * (d["flops"] / n) / rec["algorithmic_bytes_per_launch"])
                row["traffic_source"] = "profiles/%s" % os.path.basename(path)
            else:
                row["traffic_at_pmc_launch_size"] = int(rec["hbm_bytes_per_launch"])
                row["traffic_source"] = "profiles/%s (launches of another size: not rescaled)" % os.path.basename(path)
        worst = max(d["classes"].items(), key=lambda kv: kv[1][1])
        row["largest_class"] = {"class": worst[0], "launches": worst[1][2],
                                "avg_launch_us": round(worst[1][1] / worst[1][2] * 1e6, 2),
                                "achieved": round(worst[1][0] / worst[1][1] / 1e9, 1)}
        rows.append(row)
    return rows


def cpu_baseline(sample_h, sample_w):
    """CPU oracle port of the same encode+decode on one bounded frame on the host cores this
    process may use: the oracle's C kernels run their output loops under OpenMP, the dense
    convolutions are torch's CPU library kernels; transform / entropy splits as BASELINE.md
    section 2 asks, threads of every leg reported"""
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from pseudocylindrical_convolution_amd.pseudo_codec import latent_shape
    from oracle import pconv_cpu, coder_cpu
    backend.use(pconv_cpu, coder_cpu)
    pconv_cpu.set_detmath(True)
    cores = pconv_cpu.host_cpu_share()   # affinity mask cut down to the cgroup's CPU quota
    threads = max(1, min(cores, 32))     # oneDNN on these small tensors stops scaling long before 128
    omp_threads = pconv_cpu.set_num_threads(threads)
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        enc, dec = make_codec(0)
        x = synthetic_frame(sample_h, sample_w, 100, "cpu")
        path = os.path.join(tempfile.mkdtemp(), "cpu.bin")
        h, w = latent_shape(sample_h, sample_w)
        t0 = time.perf_counter()
        sym = enc.symbols(x)
        t1 = time.perf_counter()
        enc.ent.start(path)
        enc.ent(sym)
        t2 = time.perf_counter()
        dec.ent.start(path)
        back = dec.ent(2 * h, 2 * w)
        t3 = time.perf_counter()
        dec.reconstruct(back)
        t4 = time.perf_counter()
        assert torch.equal(back, enc.ent.fill(sym)), "CPU baseline: decoded symbols differ"
    finally:
        torch.set_num_threads(before)
        backend.reset()
    dt = t4 - t0
    return {"value": sample_h * sample_w / dt / 1e6, "unit": "MPix/s", "cores": max(threads, omp_threads), "kind": "port",
            "sample": "1 frame %dx%d enc+dec, %.1f s (oracle C kernels on %d OpenMP threads, arithmetic coder 1 thread, "
                      "torch CPU conv on %d threads)" % (sample_w, sample_h, dt, omp_threads, threads),
            "threads": {"analysis": threads, "entropy_encode": omp_threads, "entropy_decode": omp_threads,
                        "synthesis": threads},
            "split_s": {"analysis": round(t1 - t0, 2), "entropy_encode": round(t2 - t1, 2),
                        "entropy_decode": round(t3 - t2, 2), "synthesis": round(t4 - t3, 2)}}


def cu_masked_stream(spec, device):
    """PCONV_BENCH_CU_MASK=first:count -- a stream whose kernels run on `count` compute units from bit `first` of
    the CU mask on (experiments: the transforms on one partition of the chip, the entropy chains on the other)"""
    import ctypes
    first, count = (int(v) for v in spec.split(":"))
    mask = (ctypes.c_uint32 * 8)()
    for b in range(max(first, 0), min(first + count, 256)):
        mask[b >> 5] |= 1 << (b & 31)
    hip = ctypes.CDLL("libamdhip64.so")
    stream = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), 8, mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask: %d" % rc)
    return torch.cuda.ExternalStream(stream.value, device=device)


class CodecWorkload(object):
    """BASELINE config #5's per-GPU share: F frames encoded and decoded per step on the
    native engine; frames resident in HBM, streams in host memory"""

    name = "codec"

    def __init__(self, args, rank, local, dev):
        from pseudocylindrical_convolution_amd.engine import CodecEngine
        self.H, self.W, self.F = args.height, args.width, args.frames_per_gpu
        self.enc, self.dec = make_codec(local)
        self.codec = CodecEngine(MODEL_VALID_DIM, local, self.enc, self.dec)
        first, stride = getattr(args, "first_frame", rank * self.F), getattr(args, "frame_stride", 1)
        self.frames = torch.cat([synthetic_frame(self.H, self.W, 100 + first + i * stride, dev)
                                 for i in range(self.F)], 0)
        self.bits_first, self.bits, self.rec, self.streams = None, 0, None, None
        self.local = local

    def step(self):
        # frames of the shard are coded in lock-step; streams stay in host memory
        self.streams = self.codec.encode(self.frames)
        self.rec = self.codec.decode(self.streams, self.H, self.W)
        self.bits = sum(len(s) for s in self.streams) * 8
        if self.bits_first is None:
            self.bits_first = self.bits

    def pixels_per_step(self):
        return float(self.F) * self.H * self.W

    def check(self):
        """after the timed loop: the workload was stationary (same bits as the first timed
        step) and the decoder returned exactly the symbols the encoder coded"""
        assert self.bits == self.bits_first, "bitstream size changed during the run (%d -> %d bits)" % (
            self.bits_first, self.bits)
        sym = self.codec.symbols(self.frames)
        eng = self.codec._engine("dec", sym.shape[2], sym.shape[3], self.F)
        assert torch.equal(eng.decode(self.streams), sym), "decoded symbols differ from the encoded ones"
        from pseudocylindrical_convolution_amd.pseudo_codec import ViewportMetrics
        metrics = ViewportMetrics(self.local)
        psnr = ssim = 0.0
        for i in range(self.F):
            p, s = metrics(self.frames[i:i + 1], self.rec[i:i + 1])
            psnr += p
            ssim += s
        return {"psnr_sum": psnr, "ssim_sum": ssim}

    def describe(self):
        return "ERP %dx%d encode+decode, model-idx 3 --ssim (valid_dim 56), %d frame(s)/GPU/step" % (
            self.W, self.H, self.F)


class AnalysisWorkload(object):
    """BASELINE config #3: SphereSlice + EncoderV2 forward only"""

    name = "analysis"

    def __init__(self, args, rank, local, dev):
        self.H, self.W, self.F = args.height, args.width, args.frames_per_gpu
        self.enc, _ = make_codec(local)
        self.frames = [synthetic_frame(self.H, self.W, 100 + rank * self.F + i, dev) for i in range(self.F)]
        self.bits = 0
        self.code = None

    @torch.no_grad()
    def step(self):
        for x in self.frames:
            self.code = self.enc.encoder(self.enc.slice(x))

    def pixels_per_step(self):
        return float(self.F) * self.H * self.W

    def check(self):
        assert self.code is not None and torch.isfinite(self.code).all()
        assert tuple(self.code.shape) == (16, 192, self.H // 256, self.W // 16)
        return {}

    def describe(self):
        return "analysis transform only (SphereSlice + EncoderV2: 4x pseudo-conv stages + GDN), ERP %dx%d, %d frame(s)/GPU/step" % (
            self.W, self.H, self.F)


WORKLOADS = {"codec": CodecWorkload, "analysis": AnalysisWorkload}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=sorted(WORKLOADS), default="codec")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--frames-per-gpu", type=int, default=None,
                    help="frames each rank codes per step, in lock-step through the entropy wavefront")
    ap.add_argument("--frames-total", type=int, default=None,
                    help="strong scaling: this many frames per step over ALL ranks (BASELINE config #5: 64), "
                         "rank r takes frames r::world; overrides --frames-per-gpu")
    ap.add_argument("--prime", type=int, default=2,
                    help="untimed passes before the warm-up so that the caching allocator reaches steady state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="2048x4096",
                    help="HxW of the CPU baseline sample (default: the metric frame, ~80 s on a 16-core share; 1024x2048: ~20 s)")
    ap.add_argument("--no-check", action="store_true", help="skip the round-trip / stationarity assertions")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of the N-process path on a box with ONE GPU: every rank uses cuda:0 (gloo for the "
                         "metric reduction: RCCL refuses two ranks on one device).  The ranks share the GPU, so `value` is "
                         "not a scaling figure; it shows the multi-process path end to end on the HIP engine (launcher, "
                         "pinning, host plan by LOCAL_WORLD_SIZE, reduction) and the host-side contention of N ranks")
    ap.add_argument("--emulate-local-world", type=int, default=0, metavar="N",
                    help="ONE real rank on one GPU with the host share rank 0 of N ranks on this node would get at "
                         "best: pinned to cgroup quota / N CPUs, LOCAL_WORLD_SIZE=N for the engine's thread / spin "
                         "rules (csrc/engine.cpp).  The 1 -> N curve of the HOST side without an N-GPU node")
    args = ap.parse_args(argv)
    if args.height is None:
        args.height = 2048 if args.mode == "codec" else 1024
    if args.width is None:
        args.width = 4096 if args.mode == "codec" else 2048
    if args.frames_per_gpu is None:
        args.frames_per_gpu = 8 if args.mode == "codec" else 1
    return args


def run(args, workload_cls=None, dist_backend="nccl", device_type="cuda"):
    """one rank of the benchmark (the whole job when WORLD_SIZE is 1)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = device_type == "cuda"
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    emulate = int(getattr(args, "emulate_local_world", 0) or 0)
    if emulate:
        if world != 1:
            raise SystemExit("--emulate-local-world is a one-rank experiment (WORLD_SIZE=%d)" % world)
        local_world = emulate
        os.environ["LOCAL_WORLD_SIZE"] = str(emulate)   # the native engine sizes its host threads by it
    cores = pin_rank(local, local_world, emulate=bool(emulate))   # before any thread of this rank exists
    strong = args.frames_total is not None
    if strong:
        if args.frames_total < world:
            raise SystemExit("--frames-total %d: fewer frames than ranks (%d)" % (args.frames_total, world))
        args.first_frame = rank
        args.frame_stride = world
        args.frames_per_gpu = len(range(rank, args.frames_total, world))
    share = bool(getattr(args, "share_gpu", False)) and on_gpu
    if share:
        dist_backend = "gloo"
    local_dev = 0 if share else local
    if on_gpu:
        torch.cuda.set_device(local_dev)   # before the first HIP call of this rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=dist_backend)
    dev = "cuda:%d" % local_dev if on_gpu else "cpu"
    n_joined = dist.get_world_size() if world > 1 else 1

    if on_gpu and os.environ.get("PCONV_BENCH_CU_MASK"):
        torch.cuda.set_stream(cu_masked_stream(os.environ["PCONV_BENCH_CU_MASK"], local_dev))
    load = (workload_cls or WORKLOADS[args.mode])(args, rank, local_dev, dev)

    def fence():
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(args.prime + args.warmup):
        load.step()
    load.bits_first = None
    probe = hbm = None
    if on_gpu:
        from pseudocylindrical_convolution_amd import PCONV
        probe, hbm = ConvProbe(), ConvProbe()
        PCONV.conv_probe, PCONV.hbm_probe = probe, hbm
    fence()
    t0 = time.perf_counter()
    cpu0 = time.process_time()   # user + system time of every thread of this rank
    for _ in range(args.steps):
        load.step()
    fence()
    elapsed = time.perf_counter() - t0
    host_busy = (time.process_time() - cpu0) / max(elapsed, 1e-9)   # host cores this rank kept busy, on average
    if os.environ.get("PCONV_BENCH_THREADS") and rank == 0:
        per = {}
        for name, allowed, cpu in thread_table():
            d = per.setdefault((name, allowed), [0, 0.0])
            d[0] += 1
            d[1] += cpu
        for (name, allowed), (n, cpu) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            sys.stderr.write("[bench threads] %-18s x%-3d cpus %-12s %.2f s\n" % (name, n, allowed, cpu))
    if on_gpu:
        PCONV.conv_probe = PCONV.hbm_probe = None
    extra = {} if args.no_check else load.check()

    from pseudocylindrical_convolution_amd import sharding
    local_sums = {"pixels": load.pixels_per_step() * args.steps, "bits": float(load.bits), "frames": float(load.F)}
    local_sums.update(extra)
    totals, elapsed = sharding.reduce_metrics(local_sums, elapsed, "cpu" if share else dev)

    out = None
    if rank == 0:
        per_kernel = probe.summarise() if probe is not None else {}
        roof, table = None, []
        for kernel, d in sorted(per_kernel.items(), key=lambda kv: -kv[1]["seconds"]):
            for label, (fl, tt, n, nbytes) in sorted(d["classes"].items(), key=lambda kv: -kv[1][1]):
                # fl = direct-convolution flops (2 Cin k^2 Cout per valid output pixel); a Winograd launch
                # executes 1 / 2.25 of them: `achieved` / `frac` are executed matrix-core flops (<= peak)
                gain = winograd_gain(kernel)
                row = {"class": label, "kernel": kernel, "launches": n, "avg_launch_ms": round(tt / n * 1e3, 4),
                       "gflop_per_launch": round(fl / gain / n / 1e9, 3), "achieved": round(fl / gain / tt / 1e12, 2),
                       "frac": round(fl / gain / tt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
                if gain != 1.0:
                    row["direct_equivalent"] = round(fl / tt / 1e12, 2)
                if nbytes:
                    # the same launches against the HBM roof: input + output (+ residual / gate) once each.  The
                    # 1x1 / GDN layers move 1.1-3 KB per pixel for 37-74 KFLOP: both roofs are about as far away
                    row["hbm_achieved"] = round(nbytes / tt / 1e9, 1)
                    row["hbm_frac"] = round(nbytes / tt / 1e9 / HBM_PEAK_GBS, 4)
                    row["flop_per_byte"] = round(fl / gain / nbytes, 1)
                table.append(row)
        if per_kernel:
            kernel = max(per_kernel, key=lambda k: per_kernel[k]["seconds"])
            d = per_kernel[kernel]
            gain = winograd_gain(kernel)
            direct = d["flops"] / d["seconds"] / 1e12          # direct-convolution flops of SURVEY 8d / time
            ach = direct / gain                                # what the algorithm executes on the matrix cores
            roof = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "launches": d["launches"],
                    "avg_launch_ms": round(d["seconds"] / d["launches"] * 1e3, 4), "traffic": None}
            if gain != 1.0:
                roof["algorithm"] = ("Winograd F(4x2,3x3) on v_mfma_f32_32x32x2_f32: 24 matrix multiply-adds per 4x2 outputs "
                                     "instead of 72; achieved / frac = executed flops, direct_equivalent = x 3"
                                     if gain == 3.0 else
                                     "Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32: 16 matrix multiply-adds per 2x2 outputs "
                                     "instead of 36; achieved / frac = executed flops, direct_equivalent = x 2.25")
                roof["direct_equivalent"] = round(direct, 2)
            roof.update(pmc_evidence(kernel, d["flops"] / gain / d["launches"]))  # (executed flops, as in the table)
        conv_s = sum(v["seconds"] for v in per_kernel.values()) / max(args.steps, 1)
        frames_total = max(totals["frames"], 1.0)
        config = {"workload": load.describe(), "frames_per_gpu": load.F,
                  "parallelism": "frames sharded, no data-path collective",
                  "residency": "frames and reconstructions resident in HBM; PCIe carries CDF rows / symbols / streams only "
                               "(copying 8 frames in and out would add ~32 ms per step, DESIGN.md section 6)",
                  "tile_conv_s_per_step": round(conv_s, 4), "cores_per_rank": cores,
                  "host_cores_busy": round(host_busy, 2)}
        if strong:
            config["frames_total"] = args.frames_total
        if emulate:
            config["emulated_local_world"] = emulate
        if share:
            config["share_gpu"] = "REHEARSAL: %d ranks on ONE GPU (gloo); not a scaling figure" % n_joined
        if load.name == "codec":
            config["bpp"] = round(totals["bits"] / (frames_total * load.H * load.W), 4)
            if not args.no_check:
                config["viewport_psnr_db"] = round(totals["psnr_sum"] / frames_total, 3)
                config["viewport_ssim"] = round(totals["ssim_sum"] / frames_total, 5)
                config["roundtrip"] = "decoded symbols == encoded symbols, bits constant over the timed steps"
        metric = "ERP MPix/s enc+dec, 4096x2048 model-idx 3" if load.name == "codec" else \
            "ERP MPix/s analysis transform only, %dx%d" % (load.W, load.H)
        out = {
            "metric": metric, "value": round(totals["pixels"] / elapsed / 1e6, 4),
            "unit": "MPix/s", "n_gpus": n_joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config, "roofline": roof,
        }
        if hbm is not None and hbm.records:
            # the gather / permute kernels of the same timed steps against the HBM roof
            out["hbm"] = hbm_table(hbm.summarise())
        if load.name == "analysis" or os.environ.get("PCONV_BENCH_TABLE"):
            out["roofline_table"] = table
        if world == 1 and on_gpu and not args.no_cpu_baseline and load.name == "codec":
            sh, sw = (int(v) for v in args.cpu_sample.split("x"))
            out["cpu_baseline"] = cpu_baseline(sh, sw)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: become the launcher.  Nothing above has touched HIP
        # (importing torch does not), and the ranks are children, not an exec of this process.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    run(args)


if __name__ == "__main__":
    main()
