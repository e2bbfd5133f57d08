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
PMC_SUMMARIES = [os.path.join(ROOT, "profiles", n) for n in ("round6_bench_pmc.json", "round5_bench_pmc.json", "round4_bench_pmc.json", "round3_bench_pmc.json",
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
    if "--share-gpu" in argv:
        # REHEARSAL only (N ranks on ONE GPU): with the frame pipe's copy streams in every process the runtime's default
        # of 4 hardware queues per process made the queued decoder chains of two processes 2.3 x slower; 8 queues
        # restore round 5's figures.  One process per GPU -- the deployment -- does not care (measured at 1 / 2 / 4 / 8
        # frames per call: profiles/round6_rehearsal.txt)
        child_env.setdefault("GPU_MAX_HW_QUEUES", "8")
    if env:
        child_env.update(env)
    return subprocess.call(cmd, env=child_env)


# ----------------------------------------------------------------------------
# workload pieces
# ----------------------------------------------------------------------------
def make_codec(device_id, vd=MODEL_VALID_DIM, weights=None):
    """the codec in a FIXED state: eval mode (the quantiser's training-mode level merge,
    pseudo_quant_cuda.cu:97-143, must not fire inside a benchmark).  weights=None: seeded random
    weights, the decoder's level table tied to the encoder's; weights=DIR: the three checkpoint files
    `DIR/3_56_{encoder,decoder,ent}.pt` loaded as the reference loads its own (pseudo_codec.py:223-227,
    strict) -- e.g. the model tools/train_round6.py trains"""
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    torch.manual_seed(1234)
    enc, dec = PC.PseudoEncoder(vd, device_id).eval(), PC.PseudoDecoder(vd, device_id).eval()
    if weights:
        prex = "%s/3_%d" % (weights, vd)
        dev = next(enc.encoder.parameters()).device
        PC.load_models(enc, prex + "_encoder.pt", prex + "_ent.pt", dev)
        PC.load_models(dec, prex + "_decoder.pt", prex + "_ent.pt", dev)
        return enc.eval(), dec.eval()
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


def frame_u8(h, w, seed, content="smooth"):
    """one frame as an image file would deliver it: uint8 (h, w, 3) on the host.  content: "smooth" = the
    synthetic frame above (low-frequency waves + noise), "procedural" = the training distribution of
    tools/train_round6.py (gradients, textures, hard-edged shapes: SphereDataset.procedural_erp)"""
    if content == "procedural":
        from pseudocylindrical_convolution_amd.SphereDataset import procedural_erp
        x = procedural_erp(h, w, 7000003 + seed, 0.5 + (seed % 5) * 0.5).unsqueeze(0)
    else:
        x = synthetic_frame(h, w, seed, "cpu")
    return (x[0] * 255.0 + 0.5).clamp_(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous()


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


def cpu_baseline(sample_h, sample_w, weights=None, content="smooth"):
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
        enc, dec = make_codec(0, weights=weights)
        x = (frame_u8(sample_h, sample_w, 100, content).permute(2, 0, 1).float() / 255.0).unsqueeze(0).contiguous()
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


# Dependency chain of ONE wavefront step of the decoder: scatter -> 12 masked-conv layers -> CDF tables
# = 14 dependent launches at the measured 2.9 us launch boundary (profiles/round4_flag_chain_probe.txt,
# round5_flag_chain_owner.txt: 2.8-2.9 us at every width), + the rows' way to the host, the arithmetic decoding of
# one plane and the symbols' way back (two PCIe hops of ~2 us and ~3 us of coder for a few hundred symbols)
CHAIN_LAUNCHES = 14
LAUNCH_BOUNDARY_US = 2.9
HOST_HOP_US = 7.0


def entropy_split(phases, load, steps):
    """the entropy wavefront against its dependency-chain floor, and the four phases of a step, from events in the
    caller's stream over the timed steps (engine.CodecEngine.phase_probe)"""
    torch.cuda.synchronize()
    per = {}
    for name, e0, e1 in phases:
        per[name] = per.get(name, 0.0) + e0.elapsed_time(e1)
    per = {k: v / max(steps, 1) for k, v in per.items()}
    h2, w2 = 2 * (load.H // 256), 2 * (load.W // 16)
    nsteps = 16 * h2 + w2 + 14 - 2                         # rows + width + groups - 2 (780 at 4096x2048)
    symbols = VALID_FRACTION * 14 * 16 * h2 * w2 * load.F  # per step of the benchmark
    floor = CHAIN_LAUNCHES * LAUNCH_BOUNDARY_US + HOST_HOP_US
    ncalls = len(load.calls)
    nsteps_total = nsteps * ncalls                         # wavefront steps one after the other per benchmark step
    out = {"encode_ms": round(per.get("entropy_encode", 0.0), 2), "decode_ms": round(per.get("entropy_decode", 0.0), 2),
           "analysis_ms": round(per.get("analysis", 0.0), 2), "synthesis_ms": round(per.get("synthesis", 0.0), 2),
           "steps": nsteps, "calls_per_step": ncalls, "frames_in_lock_step": load.calls[0]["n"],
           "chain_floor_us": round(floor, 1),
           "chain_floor": "%d dependent launches x %.1f us launch boundary + %.0f us host hop (rows out, arithmetic decoding "
                          "of one plane, symbols back)" % (CHAIN_LAUNCHES, LAUNCH_BOUNDARY_US, HOST_HOP_US)}
    if per.get("entropy_decode"):
        out["us_per_step"] = round(per["entropy_decode"] * 1e3 / nsteps_total, 1)
        out["steps_per_s"] = round(nsteps_total / (per["entropy_decode"] * 1e-3), 0)
        out["floor_frac"] = round(floor / (per["entropy_decode"] * 1e3 / nsteps_total), 3)
        out["symbols_per_s"] = round(symbols / (per["entropy_decode"] * 1e-3), 0)
    if per.get("entropy_encode"):
        out["encode_symbols_per_s"] = round(symbols / (per["entropy_encode"] * 1e-3), 0)
    return out


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
    """BASELINE config #5's per-GPU share: F frames encoded and decoded per step on the native engine.

    io="host" (default): a step starts at uint8 frames in pinned host memory and ends at uint8 reconstructions
    in pinned host memory -- the two ends of the reference's flow (pseudo_codec.py:236-247, 249-268) -- over
    engine.FramePipe: upload of the next call's frames and download of the previous call's images on copy streams
    under the current call's compute, img2tensor / tensor2img arithmetic on the device.  io="resident": frames and
    reconstructions stay in HBM (rounds 1-5's figure, kept as config.value_frames_resident).  Streams stay in
    host memory either way.

    A shard of more than --max-frames-per-call frames (strong scaling: --frames-total 64 on 1 / 2 / 4 GPUs = 64 / 32 /
    16 frames per rank) is coded as several calls of at most that many frames per step: the engine's buffers (~2 GB
    per frame of a lock-step group) and the batched transform tails stay at the size the 8-frame figure was tuned
    at, and the copies of call k + 1 / k - 1 overlap call k inside the step as they do across steps."""

    name = "codec"

    def __init__(self, args, rank, local, dev, frames=None, io=None):
        from pseudocylindrical_convolution_amd.engine import CodecEngine, FramePipe
        from pseudocylindrical_convolution_amd import PCONV
        self.H, self.W = args.height, args.width
        self.F = frames if frames is not None else args.frames_per_gpu
        self.io = io or args.io
        self.enc, self.dec = make_codec(local, weights=args.weights)
        self.codec = CodecEngine(MODEL_VALID_DIM, local, self.enc, self.dec)
        first, stride = getattr(args, "first_frame", rank * self.F), getattr(args, "frame_stride", 1)
        cap = max(1, int(getattr(args, "max_frames_per_call", 8)))
        self.calls, self.pipes = [], {}
        for lo in range(0, self.F, cap):
            n = min(cap, self.F - lo)
            host = torch.stack([frame_u8(self.H, self.W, 100 + first + (lo + i) * stride, args.content)
                                for i in range(n)], 0).pin_memory()
            if n not in self.pipes and self.io == "host":
                self.pipes[n] = {"pipe": FramePipe(n, self.H, self.W, dev), "fill": 0, "use": 0}
            # the same frames resident in HBM (io="resident", and the post-run checks): img2tensor of the same bytes
            self.calls.append({"host": host, "n": n, "frames": PCONV.frames_u8_to_f32(host.to(dev)),
                               "streams": None, "rec": None, "host_rec": None, "slot": 0})
        if self.io == "host":
            self._prefetch(0)
        self.bits_first, self.bits = None, 0
        self.local = local

    def _prefetch(self, i):
        call = self.calls[i % len(self.calls)]
        st = self.pipes[call["n"]]
        st["pipe"].prefetch(call["host"], st["fill"])
        st["fill"] ^= 1

    def step(self):
        # frames of a call are coded in lock-step; streams stay in host memory
        bits = 0
        for i, call in enumerate(self.calls):
            if self.io == "host":
                st = self.pipes[call["n"]]
                slot, st["use"] = st["use"], st["use"] ^ 1
                frames = st["pipe"].take(slot)
                self._prefetch(i + 1)                       # the next call's frames cross PCIe under this call
            else:
                frames = call["frames"]
            call["streams"] = self.codec.encode(frames)
            call["rec"] = self.codec.decode(call["streams"], self.H, self.W)
            if self.io == "host":
                call["host_rec"] = st["pipe"].give(call["rec"], slot)   # ... and this call's images under the next one
                call["slot"] = slot
            bits += sum(len(s) for s in call["streams"]) * 8
        self.bits = bits
        if self.bits_first is None:
            self.bits_first = self.bits

    def pixels_per_step(self):
        return float(self.F) * self.H * self.W

    @property
    def rec(self):
        return torch.cat([c["rec"] for c in self.calls], 0) if len(self.calls) > 1 else self.calls[0]["rec"]

    def check(self):
        """after the timed loop: the workload was stationary (same bits as the first timed
        step), the decoder returned exactly the symbols the encoder coded, and the image that
        reached host memory is tensor2img of the reconstruction"""
        assert self.bits == self.bits_first, "bitstream size changed during the run (%d -> %d bits)" % (
            self.bits_first, self.bits)
        from pseudocylindrical_convolution_amd.pseudo_codec import ViewportMetrics
        metrics = ViewportMetrics(self.local)
        psnr = ssim = 0.0
        for call in self.calls:
            sym = self.codec.symbols(call["frames"])
            eng = self.codec._engine("dec", sym.shape[2], sym.shape[3], call["n"])
            assert torch.equal(eng.decode(call["streams"]), sym), "decoded symbols differ from the encoded ones"
            if self.io == "host" and call["host_rec"] is not None:
                got = self.pipes[call["n"]]["pipe"].wait(call["slot"])
                if call is self.calls[-1] or len(self.calls) <= 2:   # (earlier calls' pinned slots have been reused since)
                    want = (call["rec"] * 255.0).permute(0, 2, 3, 1).cpu().numpy().astype("uint8")   # tensor2img, pseudo_codec.py:219-221
                    assert (got.numpy() == want).all(), "the downloaded images are not tensor2img of the reconstructions"
            for i in range(call["n"]):
                p, s = metrics(call["frames"][i:i + 1], call["rec"][i:i + 1])
                psnr += p
                ssim += s
        return {"psnr_sum": psnr, "ssim_sum": ssim}

    def describe(self):
        per_call = "" if len(self.calls) == 1 else " in %d calls of <= %d" % (len(self.calls), max(c["n"] for c in self.calls))
        return "ERP %dx%d encode+decode, model-idx 3 --ssim (valid_dim 56), %d frame(s)/GPU/step%s" % (
            self.W, self.H, self.F, per_call)


class AnalysisWorkload(object):
    """BASELINE config #3: SphereSlice + EncoderV2 forward only"""

    name = "analysis"

    def __init__(self, args, rank, local, dev):
        self.H, self.W, self.F = args.height, args.width, args.frames_per_gpu
        self.enc, _ = make_codec(local, weights=args.weights)
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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=sorted(WORKLOADS), default="codec")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--frames-per-gpu", type=int, default=None,
                    help="frames each rank codes per step, in lock-step through the entropy wavefront")
    ap.add_argument("--frames-total", type=int, default=None,
                    help="strong scaling: this many frames per step over ALL ranks (BASELINE config #5: 64), "
                         "rank r takes frames r::world; overrides --frames-per-gpu")
    ap.add_argument("--max-frames-per-call", type=int, default=8,
                    help="a rank's shard is coded in calls of at most this many frames (lock-step through the entropy "
                         "wavefront, batched transform tails); 8 = the size the engine's groups were tuned at")
    ap.add_argument("--prime", type=int, default=2,
                    help="untimed passes before the warm-up so that the caching allocator reaches steady state")
    ap.add_argument("--io", choices=["host", "resident"], default="host",
                    help="host (default): every step starts at uint8 frames in pinned host memory and ends at uint8 "
                         "reconstructions there (upload / download on copy streams beside the compute); resident: frames "
                         "and reconstructions stay in HBM (the figure of rounds 1-5)")
    ap.add_argument("--weights", default=os.environ.get("PCONV_BENCH_WEIGHTS") or None,
                    help="directory with 3_56_{encoder,decoder,ent}.pt (e.g. the model of tools/train_round6.py); "
                         "default: seeded random weights")
    ap.add_argument("--content", choices=["smooth", "procedural"], default="smooth",
                    help="synthetic frames: low-frequency waves + noise, or the procedural training distribution")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two extra timed legs (frames resident in HBM; one frame per call)")
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
    device_flag = None
    if on_gpu and workload_cls is None:
        # a rank with fewer CPUs than its call has threads asks, explicitly, for sleeping runtime waits on its device
        # (the engine's own waits are blocking events either way) and reports whether the runtime took the flag
        import ctypes
        from pseudocylindrical_convolution_amd import _native
        lib = _native.hip_lib()
        plan = [ctypes.c_int(0) for _ in range(4)]
        if lib.pconv_ee_host_plan(min(args.frames_per_gpu, args.max_frames_per_call), *[ctypes.addressof(v) for v in plan]) == 0 \
                and plan[3].value:
            device_flag = lib.pconv_device_blocking_sync(1) == 1
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
    probe = hbm = phases = None
    if on_gpu:
        from pseudocylindrical_convolution_amd import PCONV
        probe, hbm = ConvProbe(), ConvProbe()
        PCONV.conv_probe, PCONV.hbm_probe = probe, hbm
        if getattr(load, "codec", None) is not None:
            phases = load.codec.phase_probe = []
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
        if phases is not None:
            load.codec.phase_probe = None
    extra = {} if args.no_check else load.check()

    def timed_leg(work, steps, warm=1):
        """`steps` more timed steps of another configuration of the same codec (every rank runs them, same fences)"""
        for _ in range(warm):
            work.step()
        fence()
        t = time.perf_counter()
        for _ in range(steps):
            work.step()
        fence()
        return (time.perf_counter() - t) / steps

    legs = {}
    if on_gpu and load.name == "codec" and not args.no_extras and workload_cls is None:
        if load.io == "host":
            # the same frames already resident in HBM, reconstructions left there (the figure of rounds 1-5)
            load.io = "resident"
            legs["resident_s"] = timed_leg(load, min(args.steps, 5))
            load.io = "host"
        # BASELINE config #4's own shape: ONE frame per call (no lock-step partner, no batched transform tails)
        one = CodecWorkload(args, rank, local_dev, dev, frames=1)
        legs["one_frame_s"] = timed_leg(one, 3, warm=2)
        del one

    from pseudocylindrical_convolution_amd import sharding
    local_sums = {"pixels": load.pixels_per_step() * args.steps, "bits": float(load.bits), "frames": float(load.F)}
    local_sums.update(extra)
    local_sums.update(legs)   # (seconds per step of the extra legs: summed over ranks, divided by n_joined below)
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
        io = getattr(load, "io", "resident")
        config = {"workload": load.describe(), "frames_per_gpu": load.F,
                  "parallelism": "frames sharded, no data-path collective",
                  "residency": ("every step starts at uint8 frames in pinned host memory and ends at uint8 reconstructions in "
                                "pinned host memory (engine.FramePipe: step k+1's upload and step k-1's download on copy "
                                "streams under step k's compute; img2tensor / tensor2img arithmetic on the device); streams "
                                "stay in host memory" if io == "host" else
                                "frames and reconstructions resident in HBM; PCIe carries CDF rows / symbols / streams only"),
                  "weights": ("trained: %s" % args.weights) if args.weights else "seeded random (transforms seed 1234, entropy randn*0.05 seed 7)",
                  "content": args.content,
                  "tile_conv_s_per_step": round(conv_s, 4), "cores_per_rank": cores,
                  "host_cores_busy": round(host_busy, 2)}
        if totals.get("resident_s", 0.0) > 0:
            config["value_frames_resident"] = round(load.pixels_per_step() * n_joined / (totals["resident_s"] / n_joined) / 1e6, 4)
        if getattr(load, "codec", None) is not None and on_gpu:
            eng = load.codec._engine("dec", 2 * (load.H // 256), 2 * (load.W // 16), load.calls[0]["n"])
            lib = eng.lib
            config["host_waits"] = "spinning (runtime default)"
            if lib.pconv_ee_wait_mode(eng.handle) == 1:
                config["host_waits"] = "sleeping (engine: blocking events; device flag hipDeviceScheduleBlockingSync: %s)" % (
                    "in effect, read back" if device_flag else ("refused by the runtime" if device_flag is False else "not asked for"))
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
        if totals.get("one_frame_s", 0.0) > 0:
            # BASELINE config #4 as written: one 4096x2048 frame per call, host to host
            out["value_one_frame"] = round(load.H * load.W / (totals["one_frame_s"] / n_joined) / 1e6, 4)
        if phases:
            out["entropy"] = entropy_split(phases, load, args.steps)
        if hbm is not None and hbm.records:
            # the gather / permute kernels of the same timed steps against the HBM roof
            out["hbm"] = hbm_table(hbm.summarise())
        if load.name == "analysis" or os.environ.get("PCONV_BENCH_TABLE"):
            out["roofline_table"] = table
        if world == 1 and on_gpu and not args.no_cpu_baseline and load.name == "codec":
            sh, sw = (int(v) for v in args.cpu_sample.split("x"))
            out["cpu_baseline"] = cpu_baseline(sh, sw, args.weights, args.content)
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
