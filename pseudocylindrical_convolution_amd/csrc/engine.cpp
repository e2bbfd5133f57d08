// Native entropy engine: the EntEncoder / EntDecoder wavefront loops of the
// reference (pseudo_codec.py:97-114, 145-160) as a C++ host loop.
//
// Arithmetic is the per-op path's (PCONV.EntropyConv2Op & co, entropy.hip): same
// causal masks, same reduction order, same CDF construction, so a stream written
// by either path decodes with the other (tests/test_gpu_engine.py).  What differs
// is orchestration and layout:
//   * one host loop in C++, per step 1 scatter + 12 layer launches + 1 table
//     launch (the per-op path: 12 halo updates, 12 convs, 5 adds, 2 extracts,
//     4 GMM launches, driven from Python);
//   * engine-private channels-last buffers whose halos are written by the
//     producer of the value they derive from (entropy_engine.hip);
//   * the frames of a call advance in lock-step inside a GROUP, one arithmetic
//     coder each; a call with several frames is split into 2 groups with
//     their own buffers, HIP stream and (in the decoder) host thread: a group's
//     step is a latency chain -- 14 back-to-back launches, host wait, arithmetic
//     decoding -- that leaves the GPU mostly idle, so the chains run side by
//     side (measured: 4 independent chains cost 25 % more time than one; beyond
//     the 4 hardware queues they start to wait for each other);
//   * only the live rows of a step cross PCIe: the decoder's table kernel writes
//     them straight into pinned host memory and its scatter kernel reads the
//     decoded symbols from there (no copy launches inside a step); the encoder
//     never waits inside the loop: tables and labels of all steps are written to
//     one device buffer in stream order, copied once and coded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <sched.h>
#include <vector>
#include "../../include/pconv_coder.h"
#include "common.h"
#include "ee_kernels.h"

namespace {

constexpr int kLayers = 12;
constexpr int kPad = 2;
constexpr int kMaxEncodeRanges = 8;  // step ranges of a group's encode (Group::enc_done)

#define HIP_TRY(expr)                                                     \
  do {                                                                    \
    hipError_t e__ = (expr);                                              \
    if (e__ != hipSuccess) {                                              \
      pconv_set_error("engine: %s: %s", #expr, hipGetErrorString(e__));   \
      return PCONV_ELAUNCH;                                               \
    }                                                                     \
  } while (0)

#define PC_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ < 0) return rc__;  \
  } while (0)

struct Window {
  int lo, len, first, nplane;
};

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

// CPUs in the affinity mask (bench.py gives every rank a slice of its own)
static int affinity_cpus() {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) != 0) return 1;
  const int n = CPU_COUNT(&set);
  return n > 0 ? n : 1;
}

// CPU quota of the cgroup in whole CPUs (0: no quota).  A GPU box shows all 256 host CPUs in the
// affinity mask of a container that owns 16 of them: the mask alone says nothing about how many
// threads can really run.  cgroup v2 `cpu.max` ("quota period" | "max period"), then v1
// cfs_quota_us / cfs_period_us; PCONV_CGROUP_CPU_MAX names another cpu.max-format file (tests).
static int cgroup_cpu_quota() {
  const char *override_path = getenv("PCONV_CGROUP_CPU_MAX");
  if (FILE *f = fopen(override_path ? override_path : "/sys/fs/cgroup/cpu.max", "r")) {
    char quota[64] = {0};
    long long period = 0;
    const int n = fscanf(f, "%63s %lld", quota, &period);
    fclose(f);
    if (n == 2 && period > 0 && strcmp(quota, "max") != 0) {
      const long long q = atoll(quota);
      if (q > 0) return (int)std::max(1LL, q / period);
    }
    return 0;
  }
  if (override_path) return 0;
  long long q = -1, period = 0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    if (fscanf(f, "%lld", &q) != 1) q = -1;
    fclose(f);
  }
  if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
    if (fscanf(f, "%lld", &period) != 1) period = 0;
    fclose(f);
  }
  return (q > 0 && period > 0) ? (int)std::max(1LL, q / period) : 0;
}

// CPUs the host threads of THIS rank can count on: the affinity mask, cut down to the rank's share
// of the cgroup quota -- quota / LOCAL_WORLD_SIZE, the ranks of a node share one container
// (test/trainDDP_Full.py:83-86,201-204: one process per GPU on one host).
static int allowed_cpus() {
  int n = affinity_cpus();
  const int quota = cgroup_cpu_quota();
  if (quota > 0) {
    int ranks = 1;
    if (const char *env = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(env));
    n = std::min(n, std::max(1, quota / ranks));
  }
  return n;
}

// A call that runs `call_threads` host threads side by side (one per frame) on fewer CPUs than that (+ the
// caller's own thread): the rank's share of the host is the scarce resource, not the GPU.  Measured on one
// MI355X with the rank pinned to 2 / 4 CPUs (profiles/round5_host_share.txt, 8 frames): four polling
// drivers + four polling workers 58-75 MPix/s; the frames of a group decoded one after the other by the
// group's driver (no workers) 74-85; at 2 CPUs two groups instead of four 80; waits that sleep instead of
// spinning: the same speed at half the CPU time (8 ranks x 2 spinning cores would exhaust a 16-CPU quota).
static bool host_constrained(int call_threads) { return call_threads + 1 > allowed_cpus(); }

// The host side of a call of `nimg` frames, decided in ONE place (pconv_ee_host_plan exports it for the tests):
struct HostPlan {
  int groups;         // lock-step groups = decoder chains = driver threads
  int group_threads;  // threads that arithmetic-decode a group's frames (the driver included); 0 = one per frame
  int queued_chain;   // 1: the queued-ahead chain, 0: the host-driven one
  int blocking_sync;  // 1: the runtime's waits sleep instead of spinning
};
static HostPlan host_plan(int nimg) {
  HostPlan p;
  const bool constrained = host_constrained(nimg);
  // two groups up to five frames, four from six on (r3, profiles/round3_decode_groups.txt); eight make the chains
  // wait for each other (4 hardware queues).  A group is a host thread: never more of them than this rank has
  // CPUs when the frames do not fit anyway
  p.groups = nimg >= 6 ? 4 : (nimg >= 2 ? 2 : 1);
  if (constrained) p.groups = std::min(p.groups, std::max(1, allowed_cpus()));
  if (const char *env = getenv("PCONV_ENGINE_GROUPS")) p.groups = atoi(env);
  p.groups = std::max(1, std::min(p.groups, nimg));
  p.group_threads = constrained ? 1 : 0;
  if (const char *env = getenv("PCONV_ENGINE_WORKERS")) p.group_threads = std::max(1, atoi(env));
  const int largest = (nimg + p.groups - 1) / p.groups;
  // queued / host-driven, measured (MI355X, 4096x2048, two groups): 94 / 103 ms for one frame, 110 / 113 for two,
  // 141 / 137 for four, 212 / 197 for eight; the queued chain keeps two host threads per group busy (one queues,
  // one polls): with fewer CPUs than frames it loses badly (8 frames on 2 CPUs: 1 956 ms against 220)
  p.queued_chain = largest < 4 && nimg < 6 && !constrained;
  if (const char *env = getenv("PCONV_ENGINE_CHAIN")) p.queued_chain = env[0] != 'h';
  p.blocking_sync = constrained;
  if (const char *env = getenv("PCONV_ENGINE_BLOCKING_SYNC")) p.blocking_sync = atoi(env) != 0;
  return p;
}

// How long an idle StepPool worker polls before it blocks, for a call that runs `call_threads`
// host threads side by side (one per frame: group drivers + their workers).  Through the whole GPU
// part of a step (2 ms covers it) only when every one of them AND the caller's own thread have a
// CPU of this rank's share; otherwise about the host part of a step, so that the GPU waits do not
// keep frames - groups cores busy that other ranks (or other threads of this one) need.
static int step_pool_spin_us(int call_threads) {
  int spin = call_threads + 1 <= allowed_cpus() ? 2000 : 60;
  if (const char *env = getenv("PCONV_ENGINE_SPIN_US")) spin = atoi(env);
  return spin;
}

// Persistent helpers for the per-step arithmetic decoding: job(i) runs for
// i = 0 (caller) .. n-1 (workers).  A step's decoding takes tens of microseconds,
// far less than creating and joining threads, so the workers poll a generation
// counter.  When the decode's threads (one per frame) have cores of their own IN THIS
// RANK'S SHARE of the host (affinity mask and cgroup quota / ranks on the node:
// allowed_cpus) they poll through the GPU part of a step as well (~180 us): a worker
// that had gone to sleep cost a futex wake-up on the critical path of EVERY step
// (measured: 8 us per step, 0.6 % of the codec).  Otherwise -- more frames than
// cores, e.g. ranks that share a host's cores -- a worker that sees nothing for 60 us
// (about the host part of a step) blocks on a condition variable, so that the GPU
// waits do not keep nimg - 1 cores per group busy.  PCONV_ENGINE_SPIN_US overrides
// either (step_pool_spin_us).
class StepPool {
 public:
  // njobs jobs per run() on `threads` threads (the caller included): thread k takes jobs k, k + threads, ...
  StepPool(int njobs, int threads, int call_threads) : njobs_(njobs), n_(std::max(1, std::min(threads, njobs))) {
    spin_us_ = step_pool_spin_us(call_threads);
    // more runnable threads than CPUs: a polling thread gives its time slice away instead of keeping the
    // thread it waits for off the core
    polite_ = call_threads + 1 > allowed_cpus();
    for (int i = 1; i < n_; i++) workers_.emplace_back([this, i] { loop(i); });
  }
  ~StepPool() {
    stop_.store(true, std::memory_order_release);
    publish();
    for (std::thread &t : workers_) t.join();
  }
  void run(const std::function<void(int)> &job) {
    if (n_ == 1) {
      for (int j = 0; j < njobs_; j++) job(j);
      return;
    }
    job_ = &job;
    done_.store(0, std::memory_order_relaxed);
    publish();
    for (int j = 0; j < njobs_; j += n_) job(j);
    while (done_.load(std::memory_order_acquire) < n_ - 1) relax();
  }

 private:
  void relax() {
    if (polite_)
      sched_yield();
    else
      cpu_relax();
  }
  void publish() {
    gen_.fetch_add(1, std::memory_order_release);
    if (sleepers_.load(std::memory_order_acquire) > 0) {
      std::lock_guard<std::mutex> lk(mu_);
      cv_.notify_all();
    }
  }
  void loop(int i) {
    int seen = 0;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      int spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        relax();
        if ((spin_us_ <= 0 || (++spins & (polite_ ? 7 : 255)) == 0) &&
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us_) {
          std::unique_lock<std::mutex> lk(mu_);
          sleepers_.fetch_add(1, std::memory_order_acq_rel);
          cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
          sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        }
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load(std::memory_order_acquire)) return;
      for (int j = i; j < njobs_; j += n_) (*job_)(j);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  int njobs_, n_;
  int spin_us_ = 60;
  bool polite_ = false;
  std::vector<std::thread> workers_;
  const std::function<void(int)> *job_ = nullptr;
  std::atomic<int> gen_{0}, done_{0}, sleepers_{0};
  std::atomic<bool> stop_{false};
  std::mutex mu_;
  std::condition_variable cv_;
};

template <typename Fn>
void for_each_image(int nimg, Fn fn) {
  if (nimg == 1) {
    fn(0);
    return;
  }
  std::vector<std::thread> pool;
  for (int i = 1; i < nimg; i++) pool.emplace_back(fn, i);
  fn(0);
  for (std::thread &t : pool) t.join();
}

// frames that advance in lock-step: buffers, stream and schedule offsets of their own
struct Group {
  int nimg = 0, first = 0;  // frames of the call: first .. first + nimg - 1
  EeGeom geom;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  hipEvent_t step_done = nullptr;      // decoder, sleeping waits: "this step's rows are on the host" (a blocking event)
  std::vector<int> enc_bounds;         // encoder: step ranges (set_encode_ranges, encode_range)
  hipEvent_t enc_done[8] = {nullptr};  // ... and "this range's rows are on the host"
  float *ctx = nullptr;             // (nimg*npart, h+4, w+4, G)
  float *act[kLayers] = {nullptr};  // (3*nimg*npart, h+2p, w+2p, 3G), persistent across steps
  // CDF rows of the group's symbols.  Packed engine (8 symbols, total 65536: the codec): 16-byte rows (uint16 c1 .. c7
  // + label, include/pconv_coder.h) in tables_d / tables_h, no label arrays; otherwise int32 rows of nlevels + 1
  // entries and int32 labels
  int32_t *tables_d = nullptr, *labels_d = nullptr;
  int32_t *tables_h = nullptr, *labels_h = nullptr;  // pinned
  float *packed_h = nullptr;                         // pinned; decoder: symbols of the previous step [img][len]
  int32_t *flags_h = nullptr;                        // pinned, coherent: decoder chain flags (entropy_engine.hip)
  int32_t *counter_d = nullptr;                      // device: finished blocks of the running table kernel
  int32_t *step_row_d = nullptr;
  std::vector<int32_t> step_row;  // first table row of each step (x nimg), +1 end
  std::vector<int32_t> sym;
  StepPool *pool = nullptr;
  bool zeroed = false;  // ctx / act hold no uninitialised memory any more (clear())
};

}  // namespace

struct pconv_entropy_engine {
  int npart, ngroup, h, w, nimg, nlevels;
  float bias, total, beta;
  int rows, nsteps, longest_plane = 0;
  std::vector<int32_t> widths, sched_start;
  int32_t *widths_d = nullptr, *order_d = nullptr, *sched_start_d = nullptr, *vh_col = nullptr;
  EePos *pos_d = nullptr;
  EeHalo *halo_d = nullptr;
  uint32_t *tap_in_d = nullptr, *tap_hid_d = nullptr;
  float *vh_wgt = nullptr;
  int32_t *pos_plane_d = nullptr;
  HostPlan plan;                  // the host side of this engine's calls, decided ONCE (pconv_ee_create)
  bool fuse_tables = false;       // decoder: last layer + table kernel as one launch (PCONV_EE_FUSE_TABLES, at creation)
  bool stepwise_encoder = false;  // debugging aid: encode step by step like the decoder
  int encode_ranges = -1;         // step ranges of a call's last group (pconv_ee_set_encode_ranges); -1: the default
  float *lw[kLayers] = {nullptr};  // engine-owned packed weights
  // matrix-core form of the encoder's hidden layers (entropy_mfma.hip): weights as MFMA fragments, the list of
  // position blocks; mfma_waves == 0: not available for this shape (the vector kernel takes every layer)
  float *lwf[kLayers] = {nullptr};
  float *lwf4[kLayers] = {nullptr};  // PCONV_EE_MFMA_FORM=4b: fragments of the four-block form (42 -> 42 layers)
  void *mfma_blocks_d = nullptr;
  int mfma_nblocks = 0, mfma_rp = 0, mfma_ct = 0, mfma_waves = 0, mfma_nt = 0;
  std::vector<int32_t> mfma_blocks_h;  // (tile, first row, first column, 0) per block
  // blocks that hold a (position, group) pair of a step range, compacted (a launch over ALL blocks whose
  // out-of-range workgroups exit at once still took 0.75 of a full launch: r5, profiles/round5_entropy_mfma_*.txt)
  std::map<std::pair<int, int>, std::pair<void *, int>> mfma_range_blocks;
  std::mutex mfma_range_mu;
  const float *lb[kLayers] = {nullptr}, *la[kLayers] = {nullptr};
  bool bound[kLayers] = {false};
  std::vector<Group> groups;
  hipEvent_t entry = nullptr;
  size_t sym_per_img = 0, max_len = 0;
  std::vector<std::vector<uint8_t>> streams;
  std::vector<pconv_coder *> coders;
  // encode in flight (pconv_ee_encode_begin / _end)
  std::thread enc_thread;
  std::atomic<int> enc_status{0};
  std::vector<std::string> enc_errors;
  std::chrono::steady_clock::time_point enc_begin;
  double enc_wait = 0, enc_coder = 0;

  // 16-byte packed rows across PCIe instead of int32 rows + labels (40 bytes per symbol) when the tables have the
  // codec's shape; PCONV_ENGINE_ROWS=int32 keeps the wide rows (A/B); the step-by-step debugging encoder needs them
  bool packed = false;
  size_t row_bytes() const { return packed ? 16 : (size_t)(nlevels + 1) * 4; }

  size_t ctx_elems(int n) const { return (size_t)n * npart * ngroup * (h + 2 * kPad) * (w + 2 * kPad); }
  size_t act_elems(int l, int n) const {
    const int p = (l == kLayers - 1) ? 0 : kPad;
    return (size_t)3 * n * npart * 3 * ngroup * (h + 2 * p) * (w + 2 * p);
  }
  int layer_cin(int l) const { return l == 0 ? ngroup : 3 * ngroup; }

  Window window(int psum) const {
    int st = psum - ngroup + 1 < 0 ? 0 : psum - ngroup + 1;
    int end = psum < rows + w - 2 ? psum + 1 : rows + w - 1;
    if (st >= end) return {0, 0, 0, 0};
    return {sched_start[st], sched_start[end] - sched_start[st], st, end - st};
  }

  int init_group(Group &g, int first, int n, const EeGeom &base) {
    g.nimg = n;
    g.first = first;
    g.geom = base;
    g.geom.nimg = n;
    // highest priority: the step kernels are short and latency-bound.  (Measured:
    // priority alone does not protect them -- running another chunk's transforms
    // beside a decode doubled its GPU waits, so CodecEngine does not overlap them.)
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // PCONV_ENGINE_CU_MASK=first:count -- the group's stream on `count` compute units from bit `first` of the CU
    // mask on (consecutive bits alternate over the XCDs and their shader engines): a partition of the chip for
    // the chains, so that they can run beside another stream's transforms (which hold whole CUs)
    int cu_first = 0, cu_count = 0;
    if (const char *m = getenv("PCONV_ENGINE_CU_MASK")) (void)sscanf(m, "%d:%d", &cu_first, &cu_count);
    if (cu_count > 0) {
      uint32_t mask[8] = {0};
      for (int b = cu_first; b < cu_first + cu_count && b < 256; b++)
        if (b >= 0) mask[b >> 5] |= 1u << (b & 31);
      HIP_TRY(hipExtStreamCreateWithCUMask(&g.stream, 8, mask));
    } else {
      HIP_TRY(hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, greatest));
    }
    // sleeping waits (host-constrained rank, host_plan): the events the host threads wait on are BLOCKING events --
    // hipEventSynchronize sleeps on the signal's interrupt instead of spinning.  A property of these events only:
    // no device-wide schedule flag is touched (torch's own synchronisations keep their behaviour)
    const unsigned evflags = hipEventDisableTiming | (plan.blocking_sync ? hipEventBlockingSync : 0u);
    HIP_TRY(hipEventCreateWithFlags(&g.done, evflags));
    HIP_TRY(hipEventCreateWithFlags(&g.step_done, evflags));
    for (int k = 0; k < kMaxEncodeRanges; k++) HIP_TRY(hipEventCreateWithFlags(&g.enc_done[k], evflags));
    g.step_row.assign(nsteps + 1, 0);
    for (int s = 0; s < nsteps; s++) g.step_row[s + 1] = g.step_row[s] + window(s).len * n;
    HIP_TRY(hipMalloc(&g.step_row_d, g.step_row.size() * 4));
    HIP_TRY(hipMemcpy(g.step_row_d, g.step_row.data(), g.step_row.size() * 4, hipMemcpyHostToDevice));
    g.geom.step_row = g.step_row_d;
    HIP_TRY(hipMalloc(&g.ctx, ctx_elems(n) * 4));
    for (int l = 0; l < kLayers; l++) HIP_TRY(hipMalloc(&g.act[l], act_elems(l, n) * 4));
    const size_t all_rows = sym_per_img * n;
    HIP_TRY(hipMalloc(&g.tables_d, all_rows * row_bytes()));
    HIP_TRY(hipHostMalloc(&g.tables_h, all_rows * row_bytes()));
    if (!packed) {
      HIP_TRY(hipMalloc(&g.labels_d, all_rows * 4));
      HIP_TRY(hipHostMalloc(&g.labels_h, all_rows * 4));
    }
    HIP_TRY(hipHostMalloc(&g.packed_h, (size_t)n * max_len * 4));
    HIP_TRY(hipHostMalloc(&g.flags_h, 64, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipMalloc(&g.counter_d, 64));
    HIP_TRY(hipMemset(g.counter_d, 0, 64));
    return PCONV_OK;
  }

  int init(const float *tile_weight) {
    rows = h * npart;
    nsteps = rows + w + ngroup - 2;
    stepwise_encoder = getenv("PCONV_ENGINE_STEPWISE_ENCODER") != nullptr;
    {
      const char *env = getenv("PCONV_ENGINE_ROWS");
      packed = nlevels == 8 && total == 65536.f && !stepwise_encoder && !(env && env[0] == 'i');
    }
    widths.assign(npart, 0);
    PC_TRY(pconv_host_tile_widths(tile_weight, npart, rows, w, widths.data()));
    std::vector<int32_t> order((size_t)rows * w);
    sched_start.assign(rows + w, 0);
    PC_TRY(pconv_host_wavefront(widths.data(), npart, h, w, order.data(), sched_start.data()));
    sym_per_img = (size_t)sched_start[rows + w - 1] * ngroup;
    for (int p = 0; p + 1 < rows + w; p++)
      if (sched_start[p + 1] - sched_start[p] > longest_plane) longest_plane = sched_start[p + 1] - sched_start[p];
    for (int s = 0; s < nsteps; s++)
      if ((size_t)window(s).len > max_len) max_len = window(s).len;
    HIP_TRY(hipMalloc(&widths_d, npart * 4));
    HIP_TRY(hipMemcpy(widths_d, widths.data(), npart * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&order_d, order.size() * 4));
    HIP_TRY(hipMemcpy(order_d, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&sched_start_d, sched_start.size() * 4));
    HIP_TRY(hipMemcpy(sched_start_d, sched_start.data(), sched_start.size() * 4, hipMemcpyHostToDevice));
    {
      const size_t n = (size_t)npart * 2 * kPad * w;
      std::vector<int32_t> col(n);
      std::vector<float> wg(n);
      PC_TRY(pconv_host_causal_table(widths.data(), npart, h, w, kPad, col.data(), wg.data()));
      HIP_TRY(hipMalloc(&vh_col, n * 4));
      HIP_TRY(hipMalloc(&vh_wgt, n * 4));
      HIP_TRY(hipMemcpy(vh_col, col.data(), n * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(vh_wgt, wg.data(), n * 4, hipMemcpyHostToDevice));
      // reverse map: interior (global row, column) -> halo entries interpolated from it
      // (entry en = ((tile*2 + side)*kPad + r)*w + column reads source columns c and
      // c+1, circular, of the neighbouring tile's row; entry columns past the tile's
      // valid width are never read and stay out).  One EeHalo record per (source, entry).
      const int win = w + 2 * kPad, hp = h + 2 * kPad;
      auto padded = [&](int tile, int prow, int pcol) { return (int32_t)((tile * hp + prow) * win + pcol); };
      std::vector<std::vector<EeHalo>> lists((size_t)rows * w);
      for (size_t en = 0; en < n; en++) {
        const int cp = (int)(en % w);
        size_t q = en / w;
        const int r = (int)(q % kPad);
        q /= kPad;
        const int side = (int)(q & 1), tg = (int)(q >> 1);
        const int c = col[en];
        if (c == -2 || cp >= widths[tg]) continue;
        const int srow = side ? (tg + 1) * h + r : tg * h - kPad + r;
        if (srow < 0 || srow >= rows) continue;
        const int st = srow / h, sr = srow - st * h;
        const int wst = widths[st];
        int c1 = c + 1;
        c1 = c1 >= wst ? c1 - wst : c1;
        EeHalo rec;
        rec.dst = padded(tg, side ? h + kPad + r : r, cp + kPad);
        rec.t = wg[en];
        const int wrap = cp < kPad ? widths[tg] : 0;
        const int32_t pa = c >= 0 ? padded(st, sr + kPad, c + kPad) : -1;
        const int32_t pb = padded(st, sr + kPad, c1 + kPad);
        if (c >= 0) {  // the producer is source a; the other one is b (or a itself when c1 == c)
          rec.other = pb;
          rec.info = wrap | (c1 == c ? (1 << 29) : 0);
          lists[(size_t)srow * w + c].push_back(rec);
        }
        if (c1 != c) {  // the producer is source b
          rec.other = pa;
          rec.info = wrap | (1 << 30);
          lists[(size_t)srow * w + c1].push_back(rec);
        }
      }
      std::vector<int32_t> rstart(lists.size() + 1, 0);
      std::vector<EeHalo> recs;
      for (size_t k = 0; k < lists.size(); k++) {
        if (lists[k].size() > 15) {
          pconv_set_error("ee_create: %zu halo entries hang on one position (record format holds 15)", lists[k].size());
          return PCONV_EINVAL;
        }
        recs.insert(recs.end(), lists[k].begin(), lists[k].end());
        rstart[k + 1] = (int32_t)recs.size();
      }
      if (recs.size() >= (1u << 27)) {
        pconv_set_error("ee_create: too many halo records");
        return PCONV_EINVAL;
      }
      if (recs.empty()) recs.push_back(EeHalo{0, -1, 0.f, 0});
      HIP_TRY(hipMalloc(&halo_d, recs.size() * sizeof(EeHalo)));
      HIP_TRY(hipMemcpy(halo_d, recs.data(), recs.size() * sizeof(EeHalo), hipMemcpyHostToDevice));
      // one EePos record per schedule entry
      const int npos = sched_start[rows + w - 1];
      std::vector<EePos> pos(npos);
      for (int i = 0; i < npos; i++) {
        const int hw = order[i], tw = hw % w, row = hw / w, tg = row / h, th = row - tg * h;
        EePos p;
        p.pix = padded(tg, th, tw);
        p.hw = hw;
        p.wrap = tw < kPad ? widths[tg] : 0;
        const int cnt = (th < kPad || th >= h - kPad) ? rstart[(size_t)row * w + tw + 1] - rstart[(size_t)row * w + tw] : 0;
        p.rev = (rstart[(size_t)row * w + tw] << 4) | cnt;
        pos[i] = p;
      }
      HIP_TRY(hipMalloc(&pos_d, pos.size() * sizeof(EePos)));
      HIP_TRY(hipMemcpy(pos_d, pos.data(), pos.size() * sizeof(EePos), hipMemcpyHostToDevice));
      // byte offset of reduction entry kk = tap*cin + ci inside a 5 x 5 x cin window
      for (int pass = 0; pass < 2; pass++) {
        const int cin = pass == 0 ? ngroup : 3 * ngroup;
        std::vector<uint32_t> tap(ee_slab_slots(cin), 0u);
        for (int kk = 0; kk < cin * 25; kk++) {
          const int t5 = kk / cin, ci = kk - t5 * cin;
          tap[kk] = 4u * (uint32_t)(((t5 / 5) * win + t5 % 5) * cin + ci);
        }
        uint32_t **dst = pass == 0 ? &tap_in_d : &tap_hid_d;
        HIP_TRY(hipMalloc(dst, tap.size() * 4));
        HIP_TRY(hipMemcpy(*dst, tap.data(), tap.size() * 4, hipMemcpyHostToDevice));
      }
    }
    EeGeom base = {npart, ngroup, h, w, nimg, widths_d, order_d, sched_start_d, vh_col, vh_wgt, pos_d,
                   halo_d, tap_in_d, tap_hid_d, nullptr, nullptr, 0};
    {  // bulk (encoder) maps
      const int npos = sched_start[rows + w - 1];
      std::vector<int32_t> pp(npos);
      for (int p = 0; p + 1 < rows + w; p++)
        for (int i = sched_start[p]; i < sched_start[p + 1]; i++) pp[i] = p;
      HIP_TRY(hipMalloc(&pos_plane_d, pp.size() * 4));
      HIP_TRY(hipMemcpy(pos_plane_d, pp.data(), pp.size() * 4, hipMemcpyHostToDevice));
      base.pos_plane = pos_plane_d;
      base.npos = npos;
    }
    for (int l = 0; l < kLayers; l++) HIP_TRY(hipMalloc(&lw[l], ee_packed_floats(3, 3 * ngroup, layer_cin(l)) * 4));
    {
      // PCONV_EE_BULK=valu keeps the vector kernel for every layer of the encoder (A/B; identical streams)
      const char *env = getenv("PCONV_EE_BULK");
      int rp = 0, ct = 0, wv = 0, nt = 0;
      // even widths only: the patch goes to LDS in 16-byte pieces and a padded row of an odd width ends on half a
      // piece (a pixel is 14 / 42 floats) -- the codec's symbol planes are Dtow(2) outputs, always even; direct
      // pconv_ee_create users with an odd w get the vector kernel
      if (!(env && env[0] == 'v') && ngroup == 14 && (w & 1) == 0 && ee_mfma_block_shape(h, 3 * ngroup, &rp, &ct, &wv, &nt)) {
        std::vector<int32_t> blk;
        for (int t = 0; t < npart; t++)
          for (int r0 = 0; r0 < h; r0 += nt * rp)
            for (int c0 = 0; c0 < widths[t]; c0 += 16 * ct) {
              const int32_t rec[4] = {t, r0, c0, 0};
              blk.insert(blk.end(), rec, rec + 4);
            }
        mfma_blocks_h = blk;
        if (!blk.empty()) {
          HIP_TRY(hipMalloc(&mfma_blocks_d, blk.size() * 4));
          HIP_TRY(hipMemcpy(mfma_blocks_d, blk.data(), blk.size() * 4, hipMemcpyHostToDevice));
          // (the input layer -- 14 channels, one context for the three sets -- only in the one-row direct form;
          // PCONV_EE_BULK0=valu keeps the vector kernel for it)
          const char *env0 = getenv("PCONV_EE_BULK0");
          const char *wsrc = getenv("PCONV_EE_MFMA_WSRC");
          const bool layer0 = nt == 1 && !(wsrc && wsrc[0] == 'r') && !(env0 && env0[0] == 'v');
          // the hidden layers: four lane classes per instruction (one row per wave, weights fetched directly) unless
          // PCONV_EE_MFMA_FORM=16x4 asks for the 16 x 16 x 4 form (round-5 A/B: profiles/round5_entropy_mfma_variants.txt)
          const char *form = getenv("PCONV_EE_MFMA_FORM");
          const bool four = nt == 1 && !(wsrc && wsrc[0] == 'r') && !(form && form[0] == '1');
          for (int l = layer0 ? 0 : 1; l < kLayers; l++)
            if (l == 0 || !four) HIP_TRY(hipMalloc(&lwf[l], (size_t)ee_mfma_packed_floats(3, layer_cin(l)) * 4));
          if (four)
            for (int l = 1; l < kLayers; l++)
              HIP_TRY(hipMalloc(&lwf4[l], (size_t)ee_mfma4_packed_floats(3, layer_cin(l)) * 4));
          mfma_nblocks = (int)(blk.size() / 4);
          mfma_rp = rp, mfma_ct = ct, mfma_waves = wv, mfma_nt = nt;
        }
      }
    }
    // decoder chains run side by side, one per group: two groups up to five frames, four from six on
    // (r3, whole codec at 8 frames, host-driven chains: 2 / 3 / 4 groups 194-199 / 183-187 / 178-180 ms per
    // decode; at 4 frames two groups of two on the queued chain stay best: profiles/round3_decode_groups.txt);
    // eight make the chains wait for each other (4 hardware queues)
    const int ngroups = plan.groups;
    groups.resize(ngroups);
    for (int k = 0, first = 0; k < ngroups; k++) {
      const int n = nimg / ngroups + (k < nimg % ngroups ? 1 : 0);
      PC_TRY(init_group(groups[k], first, n, base));
      first += n;
    }
    HIP_TRY(hipEventCreateWithFlags(&entry, hipEventDisableTiming));
    streams.resize(nimg);
    for (int i = 0; i < nimg; i++) coders.push_back(pconv_coder_new(nullptr));
    return PCONV_OK;
  }

  void release() {
    auto freed = [](void *p) {
      if (p) (void)hipFree(p);
    };
    freed(widths_d); freed(order_d); freed(sched_start_d); freed(vh_col); freed(vh_wgt);
    freed(pos_plane_d); freed(pos_d); freed(halo_d); freed(tap_in_d); freed(tap_hid_d);
    for (int l = 0; l < kLayers; l++) freed(lw[l]);
    for (int l = 0; l < kLayers; l++) freed(lwf[l]);
    for (int l = 0; l < kLayers; l++) freed(lwf4[l]);
    freed(mfma_blocks_d);
    for (auto &kv : mfma_range_blocks) freed(kv.second.first);
    for (Group &g : groups) {
      freed(g.ctx); freed(g.tables_d); freed(g.labels_d); freed(g.step_row_d);
      for (int l = 0; l < kLayers; l++) freed(g.act[l]);
      if (g.tables_h) (void)hipHostFree(g.tables_h);
      if (g.labels_h) (void)hipHostFree(g.labels_h);
      if (g.packed_h) (void)hipHostFree(g.packed_h);
      if (g.flags_h) (void)hipHostFree(g.flags_h);
      freed(g.counter_d);
      if (g.done) (void)hipEventDestroy(g.done);
      if (g.step_done) (void)hipEventDestroy(g.step_done);
      for (hipEvent_t &ev : g.enc_done)
        if (ev) (void)hipEventDestroy(ev);
      if (g.stream) (void)hipStreamDestroy(g.stream);
    }
    if (entry) (void)hipEventDestroy(entry);
    for (pconv_coder *c : coders) pconv_coder_free(c);
  }

  // "Stale but finite" no longer holds after a failed call (a launch that did not run, an aborted chain) or
  // with new weights (an overflow of the old ones would stay in the buffers: 0 * inf = nan): the next call
  // zeroes again.
  void distrust_buffers() {
    for (Group &g : groups) g.zeroed = false;
  }

  // group streams start after everything the caller has queued (weights, symbols)
  int fork(hipStream_t caller) {
    HIP_TRY(hipEventRecord(entry, caller));
    for (Group &g : groups) HIP_TRY(hipStreamWaitEvent(g.stream, entry, 0));
    return PCONV_OK;
  }
  // ... and the caller's stream continues after the groups
  int join(hipStream_t caller) {
    for (Group &g : groups) {
      HIP_TRY(hipEventRecord(g.done, g.stream));
      HIP_TRY(hipStreamWaitEvent(caller, g.done, 0));
    }
    return PCONV_OK;
  }

  // The context and activation buffers are zeroed ONCE per group (r4; 2 GB per two-frame group and call before:
  // 6 ms of the 8-frame step).  What must be zero stays zero -- dead columns, the left halo columns and the rows
  // beyond the first / last tile are never written -- and everything else is either rewritten before it is read
  // (every live position and its halo / wrap entries, by the producer of the value) or read through a zero weight
  // (the causal mask: a stale value of the previous call is finite, so fmaf(stale, 0, acc) == acc).
  int clear(Group &g) {
    HIP_TRY(hipMemsetAsync(g.counter_d, 0, 64, g.stream));  // table-kernel block counter, scatter relay word
    // PCONV_ENGINE_CLEAR_EVERY_CALL=1: the pre-r4 behaviour (every call starts from zeroed buffers)
    static const bool every_call = getenv("PCONV_ENGINE_CLEAR_EVERY_CALL") && atoi(getenv("PCONV_ENGINE_CLEAR_EVERY_CALL"));
    if (g.zeroed && !every_call) return PCONV_OK;
    HIP_TRY(hipMemsetAsync(g.ctx, 0, ctx_elems(g.nimg) * 4, g.stream));
    for (int l = 0; l < kLayers; l++) HIP_TRY(hipMemsetAsync(g.act[l], 0, act_elems(l, g.nimg) * 4, g.stream));
    g.zeroed = true;
    return PCONV_OK;
  }

  // the 12 layers of one wavefront step (EntropyConvDBT / EntropyResidualBlockDBT
  // of pseudo_codec.py:27-51, 79-87)
  int network_step(Group &g, int s, const Window &cur, int nlayers = kLayers) {
    if (cur.len <= 0) return PCONV_OK;
    const int hid = 3 * ngroup;
    for (int l = 0; l < nlayers; l++) {
      const float *in = (l == 0) ? g.ctx : g.act[l - 1];
      // second conv of a residual block: += block input, folded into the epilogue
      const float *res = (l >= 2 && l <= 10 && (l % 2) == 0) ? g.act[l - 2] : nullptr;
      PC_TRY(ee_conv(&g.geom, in, l == 0, lw[l], lb[l], la[l], res, g.act[l], layer_cin(l), hid, l == 0 ? 5 : 6,
                     l == kLayers - 1 ? 0 : kPad, cur.first, cur.nplane, longest_plane, s, g.stream));
    }
    return PCONV_OK;
  }

  // schedule entries that have a (position, group) pair in the wavefront steps [s_lo, s_hi): planes
  // s_lo - ngroup + 1 .. s_hi - 1 (the schedule is sorted by plane)
  void step_range_entries(int s_lo, int s_hi, int *first, int *count) const {
    const int nplane = rows + w - 1;
    const int p_lo = std::max(0, s_lo - (ngroup - 1)), p_hi = std::min(nplane, s_hi);
    *first = sched_start[p_lo];
    *count = p_hi > p_lo ? sched_start[p_hi] - sched_start[p_lo] : 0;
  }

  // the blocks of the matrix-core kernel that hold a pair of the steps [s_lo, s_hi) (entropy_mfma.hip's own test:
  // planes pmin .. pmax of the block, groups 0 .. ngroup - 1), as a device list; built once per range
  int mfma_blocks_of_range(int s_lo, int s_hi, const void **list, int *count) {
    if (s_lo <= 0 && s_hi >= nsteps) {
      *list = mfma_blocks_d;
      *count = mfma_nblocks;
      return PCONV_OK;
    }
    std::lock_guard<std::mutex> lk(mfma_range_mu);
    auto it = mfma_range_blocks.find({s_lo, s_hi});
    if (it == mfma_range_blocks.end()) {
      std::vector<int32_t> act;
      const int br = mfma_nt * mfma_rp, bc = 16 * mfma_ct;
      for (size_t k = 0; k + 3 < mfma_blocks_h.size(); k += 4) {
        const int t = mfma_blocks_h[k], r0 = mfma_blocks_h[k + 1], c0 = mfma_blocks_h[k + 2];
        const int cmax = std::min(c0 + bc, (int)widths[t]) - 1;
        const int pmin = t * h + r0 + c0, pmax = t * h + r0 + br - 1 + cmax;
        if (pmax + ngroup - 1 < s_lo || pmin >= s_hi) continue;
        act.insert(act.end(), mfma_blocks_h.begin() + k, mfma_blocks_h.begin() + k + 4);
      }
      void *d = nullptr;
      if (!act.empty()) {
        HIP_TRY(hipMalloc(&d, act.size() * 4));
        HIP_TRY(hipMemcpy(d, act.data(), act.size() * 4, hipMemcpyHostToDevice));
      }
      it = mfma_range_blocks.emplace(std::make_pair(s_lo, s_hi), std::make_pair(d, (int)(act.size() / 4))).first;
    }
    *list = it->second.first;
    *count = it->second.second;
    return PCONV_OK;
  }

  // every layer once over the (plane, group) pairs of the steps [s_lo, s_hi): the encoder knows all symbols.  A
  // pair's unmasked inputs lie in steps <= its own (masks of constrain 5 / 6), i.e. in this range's previous
  // layer or in an earlier range: ranges are evaluated in step order, layer by layer inside a range; what a
  // window holds of later steps is stale but finite and meets a masked (zero) weight.
  int network_bulk(Group &g, int s_lo, int s_hi) {
    const int hid = 3 * ngroup;
    int first = 0, count = 0;
    step_range_entries(s_lo, s_hi, &first, &count);
    const void *blist = nullptr;
    int nblist = 0;
    if (mfma_waves) PC_TRY(mfma_blocks_of_range(s_lo, s_hi, &blist, &nblist));
    for (int l = 0; l < kLayers; l++) {
      const float *in = (l == 0) ? g.ctx : g.act[l - 1];
      const float *res = (l >= 2 && l <= 10 && (l % 2) == 0) ? g.act[l - 2] : nullptr;
      if (mfma_waves && (lwf[l] || lwf4[l]) && nblist == 0) continue;  // (no block of this range: nothing to evaluate)
      if (mfma_waves && lwf4[l])
        PC_TRY(ee_conv_bulk_mfma4(&g.geom, blist, nblist, mfma_rp, mfma_ct, mfma_waves, in, lwf4[l], lb[l], la[l], res,
                                  g.act[l], layer_cin(l), hid, l == kLayers - 1 ? 0 : kPad, s_lo, s_hi, g.stream));
      else if (mfma_waves && lwf[l])
        PC_TRY(ee_conv_bulk_mfma(&g.geom, blist, nblist, mfma_rp, mfma_ct, mfma_waves, mfma_nt, in, l == 0, lwf[l], lb[l],
                                 la[l], res, g.act[l], layer_cin(l), hid, l == kLayers - 1 ? 0 : kPad, s_lo, s_hi,
                                 g.stream));
      else
        PC_TRY(ee_conv_bulk(&g.geom, in, l == 0, lw[l], lb[l], la[l], res, g.act[l], layer_cin(l), hid,
                            l == 0 ? 5 : 6, l == kLayers - 1 ? 0 : kPad, first, count, s_lo, s_hi, g.stream));
      if (l != kLayers - 1) PC_TRY(ee_halo_bulk(&g.geom, g.act[l], hid, 3 * g.nimg, g.stream));
    }
    return PCONV_OK;
  }

  size_t image_symbols() const { return (size_t)npart * ngroup * h * w; }

  // encoder, GPU part of one group: CDF rows and labels of all its symbols -> pinned host memory
  // encoder, GPU part of one group, first half: all symbols are known -- fill the context once; the causal masks
  // keep every step from seeing more than DInput2 would have given it
  int encode_prologue(Group &g, const float *symbols) {
    const float *sym = symbols + (size_t)g.first * image_symbols();
    PC_TRY(clear(g));
    PC_TRY(ee_fill_ctx(&g.geom, sym, g.ctx, -bias, g.stream));
    PC_TRY(ee_halo_bulk(&g.geom, g.ctx, ngroup, g.nimg, g.stream));
    return PCONV_OK;
  }

  // ... second half, step range k of g.enc_bounds (0 = b_0 < b_1 < ... = nsteps): the range's CDF rows and labels
  // -> pinned host memory, then the event that says so -- the arithmetic coder works on range k while the GPU is
  // on range k + 1 (of this group or of another one)
  int encode_range(Group &g, const float *symbols, int k) {
    const float *sym = symbols + (size_t)g.first * image_symbols();
    const int s_lo = g.enc_bounds[k], s_hi = g.enc_bounds[k + 1];
    int first = 0, count = 0;
    step_range_entries(s_lo, s_hi, &first, &count);
    PC_TRY(network_bulk(g, s_lo, s_hi));
    PC_TRY(ee_tables_bulk(&g.geom, g.act[kLayers - 1], sym, g.tables_d, g.labels_d, nlevels, bias, total, beta, first, count,
                          s_lo, s_hi, packed, g.stream));
    const size_t r0 = g.step_row[s_lo], r1 = g.step_row[s_hi];
    if (r1 > r0) {
      const size_t rb = row_bytes();
      HIP_TRY(hipMemcpyAsync((char *)g.tables_h + r0 * rb, (const char *)g.tables_d + r0 * rb, (r1 - r0) * rb,
                             hipMemcpyDeviceToHost, g.stream));
      if (!packed)
        HIP_TRY(hipMemcpyAsync(g.labels_h + r0, g.labels_d + r0, (r1 - r0) * 4, hipMemcpyDeviceToHost, g.stream));
    }
    HIP_TRY(hipEventRecord(g.enc_done[k], g.stream));
    return PCONV_OK;
  }

  // the debugging encoder: step by step like the decoder (PCONV_ENGINE_STEPWISE_ENCODER; int32 rows)
  int encode_stepwise(Group &g, const float *symbols) {
    const int cols = nlevels + 1;
    const float *sym = symbols + (size_t)g.first * image_symbols();
    for (int s = 0; s < nsteps; s++) {
      const Window cur = window(s);
      PC_TRY(network_step(g, s, cur));
      PC_TRY(ee_tables(&g.geom, g.act[kLayers - 1], sym, g.tables_d + (size_t)g.step_row[s] * cols,
                       g.labels_d + g.step_row[s], cur.lo, cur.len, s, nlevels, bias, total, beta, nullptr, nullptr, 0, 0,
                       g.stream));
    }
    const size_t row = g.step_row[nsteps];
    HIP_TRY(hipMemcpyAsync(g.tables_h, g.tables_d, row * cols * 4, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(g.labels_h, g.labels_d, row * 4, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipEventRecord(g.enc_done[0], g.stream));
    return PCONV_OK;
  }

  // step ranges of a group's encode: `nrange` ranges of about equal numbers of symbols
  void set_encode_ranges(Group &g, int nrange) const {
    nrange = std::max(1, std::min(nrange, kMaxEncodeRanges));
    g.enc_bounds.assign(1, 0);
    const long long total_rows = g.step_row[nsteps];
    for (int k = 1; k < nrange; k++) {
      const long long want = total_rows * k / nrange;
      int s = g.enc_bounds.back() + 1;
      while (s < nsteps && g.step_row[s] < want) s++;
      if (s < nsteps && s > g.enc_bounds.back()) g.enc_bounds.push_back(s);
    }
    g.enc_bounds.push_back(nsteps);
  }

  // decoder, GPU phase of step s for one group (everything is queued, nothing waits).
  // chained: the scatter kernel waits in memory for the host's symbols and the table kernel
  // publishes its rows there, so the step can be queued long before its inputs exist.
  int decode_enqueue(Group &g, int s, const Window &prev, const Window &cur, bool chained) {
    // Zero-copy both ways: the scatter kernel reads the decoded symbols from the pinned host
    // buffer and the table kernel writes its rows into pinned host memory (both are
    // device-visible), which removes two copy launches and their gaps from every step.
    int32_t *flags = chained ? g.flags_h : nullptr;
    if (s > 0)
      PC_TRY(ee_scatter(&g.geom, g.packed_h, g.ctx, prev.lo, prev.len, s - 1, -bias, flags, g.counter_d + 8, s, g.stream));
    if (cur.len > 0) {
      // PCONV_EE_FUSE_TABLES=1 (read when the engine is created): the last layer and the table kernel as one launch
      // (ee_conv_tables; the codec's shape with packed rows only).  Identical rows -- and 2-8 % SLOWER decodes
      // (profiles/round6_fused_tables.txt): off by default
      if (fuse_tables && packed && ngroup == 14 && nlevels == 8 && total == 65536.f) {
        PC_TRY(network_step(g, s, cur, kLayers - 1));
        PC_TRY(ee_conv_tables(&g.geom, g.act[kLayers - 2], lw[kLayers - 1], lb[kLayers - 1], g.tables_h,
                              layer_cin(kLayers - 1), cur.first, cur.nplane, longest_plane, s, cur.lo, cur.len, bias, total,
                              beta, g.counter_d, flags, s + 1, g.stream));
      } else {
        PC_TRY(network_step(g, s, cur));
        PC_TRY(ee_tables(&g.geom, g.act[kLayers - 1], nullptr, g.tables_h, nullptr, cur.lo, cur.len, s, nlevels, bias,
                         total, beta, g.counter_d, flags, s + 1, packed, g.stream));
      }
    }
    return PCONV_OK;
  }

  // arithmetic decoding of one step's rows (already in tables_h) for the frames of a group
  int decode_rows(Group &g, int s, const Window &cur) {
    const int cols = nlevels + 1;
    const size_t nrow = (size_t)cur.len * g.nimg;
    g.sym.resize(nrow);
    std::atomic<int> status{0};
    const std::function<void(int)> job = [&](int i) {
      const int rc =
          packed ? pconv_coder_decodes_rows16_i32(coders[g.first + i], (const uint16_t *)g.tables_h + (size_t)i * cur.len * 8,
                                                  g.sym.data() + (size_t)i * cur.len, cur.len)
                 : pconv_coder_decodes_i32(coders[g.first + i], g.tables_h + (size_t)i * cur.len * cols, nlevels,
                                           g.sym.data() + (size_t)i * cur.len, cur.len);
      if (rc < 0) status.store(rc);
      float *dst = g.packed_h + (size_t)i * cur.len;
      const int32_t *src = g.sym.data() + (size_t)i * cur.len;
      for (int k = 0; k < cur.len; k++) dst[k] = (float)src[k];
    };
    g.pool->run(job);
    if (status.load() < 0) {
      pconv_set_error("ee_decode: arithmetic decoder desynchronised at step %d", s);
      return PCONV_EINVAL;
    }
    return PCONV_OK;
  }

  // decoder, CPU phase of step s for one group, host-driven chain: wait for the stream,
  // decode, leave the symbols for the next step's scatter
  int decode_symbols(Group &g, int s, const Window &cur, double *t_wait, double *t_coder) {
    if (cur.len <= 0) return PCONV_OK;
    const auto t0 = std::chrono::steady_clock::now();
    // (polling a flag the table kernel publishes behind its rows instead -- the queued chain's protocol -- is
    // slower here: 183-192 vs 174-180 ms for the 8-frame decode, profiles/round4_decode_spin_poll.txt)
    if (plan.blocking_sync) {
      HIP_TRY(hipEventRecord(g.step_done, g.stream));
      HIP_TRY(hipEventSynchronize(g.step_done));
    } else {
      HIP_TRY(hipStreamSynchronize(g.stream));
    }
    const auto t1 = std::chrono::steady_clock::now();
    PC_TRY(decode_rows(g, s, cur));
    const auto t2 = std::chrono::steady_clock::now();
    *t_wait += std::chrono::duration<double>(t1 - t0).count();
    *t_coder += std::chrono::duration<double>(t2 - t1).count();
    return PCONV_OK;
  }

  // the same for the queued-ahead chain: the table kernel announces its rows in flags[1]
  // and the next scatter kernel waits for flags[0]; `failed`: the queueing thread gave up
  int decode_symbols_chained(Group &g, int s, const Window &cur, const std::atomic<int> &failed, double *t_wait,
                             double *t_coder) {
    volatile int32_t *flags = g.flags_h;
    int rc = PCONV_OK;
    if (cur.len > 0) {
      const auto t0 = std::chrono::steady_clock::now();
      long long spins = 0;
      while (__atomic_load_n(&flags[1], __ATOMIC_ACQUIRE) < s + 1) {
        cpu_relax();
        if ((++spins & 0xfffff) == 0) {
          if (failed.load(std::memory_order_acquire) || __atomic_load_n(&flags[2], __ATOMIC_ACQUIRE) ||
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
            pconv_set_error("ee_decode: the GPU chain stopped before step %d", s);
            return PCONV_ELAUNCH;
          }
        }
      }
      const auto t1 = std::chrono::steady_clock::now();
      rc = decode_rows(g, s, cur);
      const auto t2 = std::chrono::steady_clock::now();
      *t_wait += std::chrono::duration<double>(t1 - t0).count();
      *t_coder += std::chrono::duration<double>(t2 - t1).count();
    }
    // symbols of step s are in packed_h (x86 stores stay in order; the GPU reads coherent memory)
    if (rc >= 0) __atomic_store_n(&flags[0], s + 1, __ATOMIC_RELEASE);
    return rc;
  }
};

extern "C" {

pconv_entropy_engine *pconv_ee_create(int npart, int ngroup, int h, int w, int nimg, const float *tile_weight,
                                      float bias, int nlevels, float total, float beta) {
  if (npart <= 0 || ngroup <= 0 || h <= 0 || w <= 0 || nimg <= 0 || !tile_weight || nlevels <= 0) {
    pconv_set_error("ee_create: bad argument");
    return nullptr;
  }
  pconv_entropy_engine *e = new pconv_entropy_engine();
  // The host side of every call of this engine -- groups, threads per group, chain, sleeping or spinning waits --
  // is decided here, once, from the rank's share of the host as it is NOW; decode reuses it (a changed affinity
  // mask or LOCAL_WORLD_SIZE between create and decode cannot disagree with the groups that exist).
  e->plan = host_plan(nimg);
  if (const char *env = getenv("PCONV_EE_FUSE_TABLES")) e->fuse_tables = atoi(env) != 0;
  e->npart = npart; e->ngroup = ngroup; e->h = h; e->w = w; e->nimg = nimg;
  e->bias = bias; e->nlevels = nlevels; e->total = total; e->beta = beta;
  if (e->init(tile_weight) != PCONV_OK) {
    e->release();
    delete e;
    return nullptr;
  }
  return e;
}

// EXPLICIT opt-in of the application (never a side effect of creating an engine): every wait of the HIP runtime on the
// current device sleeps on an interrupt instead of spinning -- hipSetDeviceFlags(hipDeviceScheduleBlockingSync), a
// process-wide flag that also changes the framework's own synchronisations.  For ranks with fewer CPUs than threads
// (bench.py calls it when pconv_ee_host_plan says blocking_sync): 1.2 instead of 2.3 busy cores and +2 % throughput at
// 2 CPUs per rank over the engine's blocking events alone (profiles/round6_rehearsal.txt).  Returns 1 when the flag is
// in effect afterwards (read back with hipGetDeviceFlags: a runtime may refuse it on a live context), 0 when not.
int pconv_device_blocking_sync(int enable) {
  unsigned flags = 0;
  if (enable) {
    if (hipGetDeviceFlags(&flags) != hipSuccess) flags = 0;
    if (hipSetDeviceFlags((flags & ~(unsigned)hipDeviceScheduleMask) | hipDeviceScheduleBlockingSync) != hipSuccess)
      (void)hipGetLastError();  // refused: the read-back below reports it, no stale error stays behind
  }
  if (hipGetDeviceFlags(&flags) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return (flags & hipDeviceScheduleMask) == hipDeviceScheduleBlockingSync ? 1 : 0;
}

// A plain non-blocking HIP stream for the host side of the path (engine.FramePipe's copy streams): created here, not
// taken from torch's stream pool -- the first stream a process takes from that pool creates the pool's 32 + 32
// streams, and two ranks that share a GPU then lost a quarter of their throughput (r6, profiles/round6_rehearsal.txt).
int pconv_stream_create(void **stream) {
  PCONV_REQUIRE(stream, "stream_create: null pointer");
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void *)s;
  return PCONV_OK;
}

int pconv_stream_destroy(void *stream) {
  if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return PCONV_OK;
}

void pconv_ee_destroy(pconv_entropy_engine *e) {
  if (!e) return;
  if (e->enc_thread.joinable()) e->enc_thread.join();
  e->release();
  delete e;
}

// device pointers in the reference's parameter layout: weight (3, 3G, cin, 5, 5),
// bias (3, 3G), slope (3, 3G) or NULL (EntropyContextNew.py:245-249).  The weight
// is re-packed into the engine's reduction order; bias and slope are borrowed.
int pconv_ee_set_layer(pconv_entropy_engine *e, int layer, const float *weight, const float *bias,
                       const float *slope, void *stream) {
  PCONV_REQUIRE(e && layer >= 0 && layer < kLayers && weight && bias, "ee_set_layer: bad argument");
  PC_TRY(ee_pack_weight(weight, e->lw[layer], 3, 3 * e->ngroup, e->layer_cin(layer), e->ngroup, layer == 0 ? 5 : 6,
                        stream));
  if (e->lwf[layer])
    PC_TRY(ee_pack_weight_mfma(weight, e->lwf[layer], 3, 3 * e->ngroup, e->layer_cin(layer), e->ngroup, layer == 0 ? 5 : 6,
                               stream));
  if (e->lwf4[layer])
    PC_TRY(ee_pack_weight_mfma4(weight, e->lwf4[layer], 3, 3 * e->ngroup, e->layer_cin(layer), e->ngroup, 6, stream));
  e->lb[layer] = bias;
  e->la[layer] = slope;
  e->bound[layer] = true;
  e->distrust_buffers();
  return PCONV_OK;
}

// host-side sizing of the engine, exported so that it can be checked without a GPU (tests/test_host_share.py)
int pconv_ee_host_cpus(void) { return allowed_cpus(); }
int pconv_ee_spin_us(int call_threads) { return step_pool_spin_us(call_threads); }
int pconv_ee_host_plan(int nimg, int *groups, int *group_threads, int *queued_chain, int *blocking_sync) {
  PCONV_REQUIRE(nimg > 0, "ee_host_plan: bad argument");
  const HostPlan p = host_plan(nimg);
  if (groups) *groups = p.groups;
  if (group_threads) *group_threads = p.group_threads;
  if (queued_chain) *queued_chain = p.queued_chain;
  if (blocking_sync) *blocking_sync = p.blocking_sync;
  return PCONV_OK;
}

int pconv_ee_wait_mode(const pconv_entropy_engine *e) {
  PCONV_REQUIRE(e, "ee_wait_mode: null engine");
  return e->plan.blocking_sync ? 1 : 0;
}

// step ranges the LAST group of the following encode calls is evaluated in (1: one piece; 0: back to the default,
// 4 or PCONV_ENGINE_ENCODE_RANGES).  Ranges buy an earlier start of the arithmetic coder at the price of a second
// evaluation of the blocks on the range boundaries: worth it only for the encode whose coding nothing else hides
// (the last chunk of a pipelined CodecEngine.encode).
int pconv_ee_set_encode_ranges(pconv_entropy_engine *e, int nrange) {
  PCONV_REQUIRE(e && nrange >= 0 && nrange <= kMaxEncodeRanges, "ee_set_encode_ranges: bad argument");
  e->encode_ranges = nrange > 0 ? nrange : -1;
  return PCONV_OK;
}

long long pconv_ee_symbols_per_image(const pconv_entropy_engine *e) { return e ? (long long)e->sym_per_img : -1; }
int pconv_ee_steps(const pconv_entropy_engine *e) { return e ? e->nsteps : -1; }

// symbols: device float (nimg*npart, ngroup, h, w) quantiser indices (dead
// columns are ignored).  Streams are kept inside the engine (pconv_ee_stream).
// encode in two halves.  begin: the GPU part of every group is queued behind the caller's
// stream and a host thread is started that, group by group, waits for the tables and runs the
// arithmetic coders (one thread per frame); begin returns at once, so the caller can go on
// queueing GPU work (the analysis transform of the next frames) while the CPU codes.  end:
// joins that thread, makes the caller's stream wait for the group streams and reports.  The
// symbols tensor must stay alive until end.
int pconv_ee_encode_begin(pconv_entropy_engine *e, const float *symbols, void *stream) {
  PCONV_REQUIRE(e && symbols, "ee_encode: bad argument");
  PCONV_REQUIRE(!e->enc_thread.joinable(), "ee_encode: the previous encode has not been ended");
  for (int l = 0; l < kLayers; l++) PCONV_REQUIRE(e->bound[l], "ee_encode: layer %d has no weights", l);
  hipStream_t caller = as_stream(stream);
  // The bulk kernels fill the GPU on their own, so the groups run one after the other (not
  // side by side like the decoder's latency-bound steps) and IN the caller's stream: queued on
  // streams of their own they are starved by whatever the caller queues next (measured: behind
  // the analysis transform of the following frames the tables arrived 0.4-0.5 s late).  The
  // first group's tables are on the host half-way and its frames are coded on the CPU while
  // the GPU works on the second group and on what the caller queues after this call.
  // PCONV_ENGINE_ENCODE_RANGES: step ranges of the call's last group (default 4; 1 = as one piece)
  const int last_ranges = e->encode_ranges > 0 ? e->encode_ranges
                          : (getenv("PCONV_ENGINE_ENCODE_RANGES") ? atoi(getenv("PCONV_ENGINE_ENCODE_RANGES")) : 4);
  // A call whose coding nothing hides (ranges asked for) takes ALL its groups through the ranges together, range by
  // range: every frame's coder starts after the first range of its group, and the last frame's GPU work no longer
  // ends a whole frame's coding before its coder does (r5: the tail behind an 8-frame encode 9 -> 3 ms).
  // PCONV_ENGINE_ENCODE_INTERLEAVE=0: group by group, ranges on the last group only (the round-4 order).
  const bool interleave = last_ranges > 1 && e->groups.size() > 1 && !e->stepwise_encoder &&
                          !(getenv("PCONV_ENGINE_ENCODE_INTERLEAVE") && atoi(getenv("PCONV_ENGINE_ENCODE_INTERLEAVE")) == 0);
  {
    std::vector<hipStream_t> own;
    for (Group &g : e->groups) {
      own.push_back(g.stream);
      g.stream = caller;
    }
    int rc = PCONV_OK;
    const size_t ng = e->groups.size();
    if (e->stepwise_encoder) {
      for (size_t k = 0; k < ng && rc >= 0; k++) {
        e->set_encode_ranges(e->groups[k], 1);
        rc = e->encode_prologue(e->groups[k], symbols);
        if (rc >= 0) rc = e->encode_stepwise(e->groups[k], symbols);
      }
    } else if (interleave) {
      size_t most = 0;
      for (size_t k = 0; k < ng && rc >= 0; k++) {
        e->set_encode_ranges(e->groups[k], last_ranges);
        most = std::max(most, e->groups[k].enc_bounds.size() - 1);
        rc = e->encode_prologue(e->groups[k], symbols);
      }
      for (size_t r = 0; r < most && rc >= 0; r++)
        for (size_t k = 0; k < ng && rc >= 0; k++)
          if (r + 1 < e->groups[k].enc_bounds.size()) rc = e->encode_range(e->groups[k], symbols, (int)r);
    } else {
      for (size_t k = 0; k < ng && rc >= 0; k++) {
        Group &g = e->groups[k];
        e->set_encode_ranges(g, k + 1 == ng ? last_ranges : 1);
        rc = e->encode_prologue(g, symbols);
        for (size_t r = 0; r + 1 < g.enc_bounds.size() && rc >= 0; r++) rc = e->encode_range(g, symbols, (int)r);
      }
    }
    for (size_t k = 0; k < ng; k++) e->groups[k].stream = own[k];
    if (rc < 0) e->distrust_buffers();
    PC_TRY(rc);
  }
  e->enc_status.store(0);
  e->enc_errors.assign(e->nimg, std::string());
  e->enc_begin = std::chrono::steady_clock::now();
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  e->enc_thread = std::thread([e, device] {
    const int cols = e->nlevels + 1;
    if (hipSetDevice(device) != hipSuccess) {
      e->enc_status.store(PCONV_ELAUNCH);
      e->enc_errors[0] = "ee_encode: hipSetDevice failed in the coder thread";
      return;
    }
    // one host thread per group: each waits for its own tables and codes its frames, so the
    // groups' coding overlaps (the GPU parts run one after the other in the caller's stream)
    std::vector<double> waits(e->groups.size(), 0.0), coders(e->groups.size(), 0.0);
    auto code_group = [&](size_t k) {
      Group &g = e->groups[k];
      if (hipSetDevice(device) != hipSuccess) {
        e->enc_status.store(PCONV_ELAUNCH);
        e->enc_errors[g.first] = "ee_encode: hipSetDevice failed in a coder thread";
        return;
      }
      const auto t0 = std::chrono::steady_clock::now();
      // the rows of the group's first step range are on the host (the frames' threads wait for the later ranges
      // themselves, each in front of the range's first step)
      if (hipEventSynchronize(g.enc_done[0]) != hipSuccess) {
        e->enc_status.store(PCONV_ELAUNCH);
        e->enc_errors[g.first] = "ee_encode: the encode stream failed";
        return;
      }
      const auto t1 = std::chrono::steady_clock::now();
      waits[k] = std::chrono::duration<double>(t1 - t0).count();
      for_each_image(g.nimg, [&](int i) {
        const int img = g.first + i;
        pconv_coder *c = e->coders[img];
        if (hipSetDevice(device) != hipSuccess) {
          e->enc_status.store(PCONV_ELAUNCH);
          e->enc_errors[img] = "ee_encode: hipSetDevice failed in a coder thread";
          return;
        }
        int rc = pconv_coder_start_encoder(c);
        size_t range = 0;
        for (int s = 0; s < e->nsteps && rc >= 0; s++) {
          if (range + 1 < g.enc_bounds.size() - 1 && s == g.enc_bounds[range + 1]) {
            range++;
            if (hipEventSynchronize(g.enc_done[range]) != hipSuccess) {
              e->enc_status.store(PCONV_ELAUNCH);
              e->enc_errors[img] = "ee_encode: the encode stream failed";
              return;
            }
          }
          const size_t len = (size_t)(g.step_row[s + 1] - g.step_row[s]) / g.nimg;
          if (!len) continue;
          const size_t r0 = (size_t)g.step_row[s] + (size_t)i * len;
          rc = e->packed ? pconv_coder_encodes_rows16(c, (const uint16_t *)g.tables_h + r0 * 8, (int)len)
                         : pconv_coder_encodes(c, g.tables_h + r0 * cols, e->nlevels, g.labels_h + r0, (int)len);
        }
        if (rc >= 0) rc = pconv_coder_end_encoder(c);
        if (rc < 0) {
          // per-image strings: the thread-local error slot belongs to the caller of end
          e->enc_status.store(rc, std::memory_order_relaxed);
          e->enc_errors[img] = std::string("ee_encode: coder of image ") + std::to_string(img) + ": " + pconv_coder_error(c);
          return;
        }
        size_t nb = 0;
        const uint8_t *p = pconv_coder_bytes(c, &nb);
        e->streams[img].assign(p, p + nb);
      });
      coders[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    };
    {
      std::vector<std::thread> pool;
      for (size_t k = 1; k < e->groups.size(); k++) pool.emplace_back(code_group, k);
      code_group(0);
      for (std::thread &t : pool) t.join();
    }
    double t_wait = 0, t_coder = 0;
    for (size_t k = 0; k < e->groups.size(); k++) {
      t_wait = std::max(t_wait, waits[k]);
      t_coder = std::max(t_coder, coders[k]);
    }
    e->enc_wait = t_wait;
    e->enc_coder = t_coder;
  });
  return PCONV_OK;
}

int pconv_ee_encode_end(pconv_entropy_engine *e, void *stream) {
  PCONV_REQUIRE(e, "ee_encode: bad argument");
  PCONV_REQUIRE(e->enc_thread.joinable(), "ee_encode_end: no encode in flight");
  e->enc_thread.join();
  if (getenv("PCONV_ENGINE_TIMING"))
    fprintf(stderr, "[pconv engine] encode %d frame(s) in %d group(s): %.1f ms, of which GPU wait %.1f ms, coder %.1f ms\n",
            e->nimg, (int)e->groups.size(),
            std::chrono::duration<double>(std::chrono::steady_clock::now() - e->enc_begin).count() * 1e3,
            e->enc_wait * 1e3, e->enc_coder * 1e3);
  (void)stream;  // everything was queued in the caller's stream: nothing to join
  if (e->enc_status.load() < 0) {
    e->distrust_buffers();
    for (const std::string &m : e->enc_errors)
      if (!m.empty()) {
        pconv_set_error("%s", m.c_str());
        break;
      }
    return e->enc_status.load() == PCONV_ELAUNCH ? PCONV_ELAUNCH : PCONV_EINVAL;
  }
  return PCONV_OK;
}

// symbols: device float (nimg*npart, ngroup, h, w) quantiser indices (dead
// columns are ignored).  Streams are kept inside the engine (pconv_ee_stream).
int pconv_ee_encode(pconv_entropy_engine *e, const float *symbols, void *stream) {
  PC_TRY(pconv_ee_encode_begin(e, symbols, stream));
  return pconv_ee_encode_end(e, stream);
}

const uint8_t *pconv_ee_stream(const pconv_entropy_engine *e, int img, size_t *nbytes) {
  if (!e || img < 0 || img >= e->nimg) return nullptr;
  if (nbytes) *nbytes = e->streams[img].size();
  return e->streams[img].data();
}

// streams[img] / nbytes[img]: host buffers.  symbols_out: device float
// (nimg*npart, ngroup, h, w) = decoded indices, zeros in the dead columns.
int pconv_ee_decode(pconv_entropy_engine *e, const uint8_t *const *streams, const size_t *nbytes,
                    float *symbols_out, void *stream) {
  PCONV_REQUIRE(e && streams && nbytes && symbols_out, "ee_decode: bad argument");
  for (int l = 0; l < kLayers; l++) PCONV_REQUIRE(e->bound[l], "ee_decode: layer %d has no weights", l);
  hipStream_t caller = as_stream(stream);
  for (int i = 0; i < e->nimg; i++)
    if (pconv_coder_start_decoder_mem(e->coders[i], streams[i], nbytes[i]) < 0) {
      pconv_set_error("ee_decode: cannot start decoder %d", i);
      return PCONV_EINVAL;
    }
  // PCONV_ENGINE_TIMING=1: where the decoder's wall time goes, printed once per call
  const bool timing = getenv("PCONV_ENGINE_TIMING") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  PC_TRY(e->fork(caller));
  // A group's step is a chain -- scatter, 12 layers, tables on the GPU, then arithmetic
  // decoding on the CPU -- of which the GPU part is latency-bound and the host part serial,
  // so the chains of the groups run side by side: every group has its own stream, pinned
  // buffers and coders, and nothing orders one group's step against another's.
  //   queued chain (default): per group one thread queues ALL steps ahead -- each step's
  //     scatter kernel waits in memory for the symbols, its table kernel announces the rows
  //     there -- and a second thread only polls, decodes and publishes: no launch and no
  //     stream synchronisation on the critical path of a step;
  //   host-driven chain: the host launches a step after it has decoded the previous one (also the
  //     checker of the queued chain: tests/test_gpu_engine.py).
  //   Which one: the queued chain moves the CDF rows and the symbols by zero-copy accesses of the kernels
  //     themselves, the host-driven one by bulk copies; measured (MI355X, 4096x2048, two groups) queued /
  //     host-driven: 94 / 103 ms for one frame, 110 / 113 for two, 141 / 137 for four, 212 / 197 for eight
  //     (four frames per group: 276 KB of rows per step); r3: from six frames on four groups, host-driven
  //     (8 frames: 178-180 ms; four groups on the queued chain 206-209).  PCONV_ENGINE_CHAIN=queued|host overrides.
  const HostPlan plan = e->plan;
  const int ng = (int)e->groups.size();
  const bool chained = plan.queued_chain != 0;
  std::vector<int> rcs(ng, PCONV_OK);
  std::vector<std::string> errors(ng);
  std::vector<double> waits(ng, 0.0), coders(ng, 0.0);
  std::vector<std::atomic<int>> failed(ng);
  for (auto &f : failed) f.store(0);
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  auto fail = [&](int k, int rc) {  // first error of a group wins; the message lives in the failing thread
    if (rcs[k] >= 0) {
      rcs[k] = rc;
      errors[k] = pconv_last_error();
    }
  };
  // queues every launch of group k's decode (queued chain)
  auto queue_all = [&](int k) {
    if (hipSetDevice(device) != hipSuccess) {
      pconv_set_error("ee_decode: hipSetDevice failed in a group thread");
      failed[k].store(1, std::memory_order_release);
      return PCONV_ELAUNCH;
    }
    Group &g = e->groups[k];
    int rc = e->clear(g);
    Window prev = {0, 0, 0, 0};
    for (int s = 0; s < e->nsteps && rc >= 0; s++) {
      const Window cur = e->window(s);
      rc = e->decode_enqueue(g, s, prev, cur, true);
      prev = cur;
    }
    // the symbols of the last step have not been scattered by a following step yet
    if (rc >= 0)
      rc = ee_scatter(&g.geom, g.packed_h, g.ctx, prev.lo, prev.len, e->nsteps - 1, -e->bias, g.flags_h,
                      g.counter_d + 8, e->nsteps, g.stream);
    if (rc >= 0)
      rc = ee_read_symbols(&g.geom, g.ctx, symbols_out + (size_t)g.first * e->image_symbols(), e->bias, g.stream);
    if (rc < 0) failed[k].store(1, std::memory_order_release);
    return rc;
  };
  auto drive = [&](int k) {
    // a new thread starts on device 0: bind it to the engine's GPU (ranks of a
    // multi-GPU job each drive their own device)
    if (hipSetDevice(device) != hipSuccess) {
      pconv_set_error("ee_decode: hipSetDevice failed in a group driver");
      fail(k, PCONV_ELAUNCH);
      return;
    }
    Group &g = e->groups[k];
    // one thread per frame of the call, unless the frames do not have CPUs of their own (host_plan: then the
    // driver decodes its frames one after the other) or PCONV_ENGINE_WORKERS says otherwise
    const int cap = plan.group_threads > 0 ? plan.group_threads : g.nimg;
    StepPool pool(g.nimg, cap, e->nimg);
    g.pool = &pool;
    int rc = PCONV_OK;
    if (chained) {
      g.flags_h[0] = g.flags_h[1] = g.flags_h[2] = 0;
      int qrc = PCONV_OK;
      std::string qerr;
      std::thread queuer([&] {
        qrc = queue_all(k);
        if (qrc < 0) qerr = pconv_last_error();
      });
      for (int s = 0; s < e->nsteps && rc >= 0; s++)
        rc = e->decode_symbols_chained(g, s, e->window(s), failed[k], &waits[k], &coders[k]);
      if (rc < 0) {
        fail(k, rc);
        // let the queued kernels run out: they only wait for this flag
        __atomic_store_n(&g.flags_h[0], 0x7fffffff, __ATOMIC_RELEASE);
      }
      queuer.join();
      if (qrc < 0 && rcs[k] >= 0) {
        rcs[k] = qrc;
        errors[k] = qerr;
      }
      if (hipStreamSynchronize(g.stream) != hipSuccess && rcs[k] >= 0) {
        rcs[k] = PCONV_ELAUNCH;
        errors[k] = "ee_decode: the decode stream failed";
      }
      if (g.flags_h[2] && rcs[k] >= 0) {
        rcs[k] = PCONV_ELAUNCH;
        errors[k] = "ee_decode: a scatter kernel timed out waiting for symbols";
      }
    } else {
      rc = e->clear(g);
      Window prev = {0, 0, 0, 0};
      for (int s = 0; s < e->nsteps && rc >= 0; s++) {
        const Window cur = e->window(s);
        rc = e->decode_enqueue(g, s, prev, cur, false);
        if (rc >= 0) rc = e->decode_symbols(g, s, cur, &waits[k], &coders[k]);
        prev = cur;
      }
      if (rc >= 0)
        rc = ee_scatter(&g.geom, g.packed_h, g.ctx, prev.lo, prev.len, e->nsteps - 1, -e->bias, nullptr, nullptr, 0,
                        g.stream);
      if (rc >= 0)
        rc = ee_read_symbols(&g.geom, g.ctx, symbols_out + (size_t)g.first * e->image_symbols(), e->bias, g.stream);
      if (rc < 0) {
        fail(k, rc);
        (void)hipStreamSynchronize(g.stream);
      }
    }
    g.pool = nullptr;
  };
  {
    std::vector<std::thread> drivers;
    for (int k = 1; k < ng; k++) drivers.emplace_back(drive, k);
    drive(0);
    for (std::thread &t : drivers) t.join();
  }
  for (int k = 0; k < ng; k++)
    if (rcs[k] < 0) {
      pconv_set_error("%s", errors[k].c_str());
      for (Group &g : e->groups) (void)hipStreamSynchronize(g.stream);
      e->distrust_buffers();
      return rcs[k];
    }
  PC_TRY(e->join(caller));
  if (timing) {
    const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    double w = 0, c = 0;
    for (int k = 0; k < ng; k++) {
      w += waits[k] / ng;
      c += coders[k] / ng;
    }
    fprintf(stderr,
            "[pconv engine] decode %d frame(s) in %d group(s), %d steps: %.1f ms; per group driver: GPU wait %.1f ms, "
            "coder %.1f ms\n",
            e->nimg, ng, e->nsteps, all * 1e3, w * 1e3, c * 1e3);
  }
  return PCONV_OK;
}

}  // extern "C"
