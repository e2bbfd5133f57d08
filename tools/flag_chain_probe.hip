// What does a dependency between two "layers" of the entropy decoder cost when it is NOT a kernel
// launch?  (SURVEY 8f-1 / DESIGN 5.)
//
// A step of the decoder is LAYERS dependent stages; stage l of step s needs stage l-1 of step s (stage 0:
// the last stage of step s-1) -- but only ITS NEIGHBOURHOOD of it: a position's 5 x 5 window reads what the
// workgroups of the parts next to it wrote.  STEPS x LAYERS stages of PARTS parts each, every part doing the
// same token work (read the three values its own part and its two neighbours wrote in the previous stage,
// add, write; optionally spin `work` cycles), four ways:
//
//   launch    one kernel launch per stage, in one stream (what the engine does)
//   p2p       ONE persistent kernel; part p of stage l waits for the flags of parts p-1, p, p+1 of the
//             previous stage only -- every flag a monotonic step count on a 64-byte line of its own, written
//             by one lane after plain stores + an agent-scope release fence + s_waitcnt, polled with relaxed
//             loads and s_sleep back-off, one agent-scope acquire fence after the three have matched
//             (MI355X_MICROARCH.md, "Valid forms").  No counter is shared by more than three pollers.
//   central   the round-3 probe: one counter per layer that every part of the stage bumps and every part
//             of the next stage polls (a central barrier, NOT point-to-point -- the round-3 write-up called
//             it "flags")
//   p2p1x     p2p with every workgroup on ONE XCD (blocks are dealt to the XCDs round-robin: the grid is
//             8 x larger and only blocks with index % 8 == 0 take part)
//
//   owner     (r5) ONE resident workgroup per PART that walks all the layers of all the steps itself -- `parts`
//             workgroups instead of layers x parts (1 300 fit the chip: 8 workgroups of 256 threads and 17 KB of
//             LDS per CU) -- stage g of part p waits for the flags of parts p-1 and p+1 only (one monotonic stage
//             count per part, a 64-byte line each).  With `slab` bytes > 0 every stage also stages a slab of that
//             size in LDS by LDS-DMA (the decoder's 17 KB weight slab of the (layer, group)), requested right
//             after the previous stage's arithmetic, i.e. BEFORE the flag wait; `launch` then stages the same
//             slab at the start of every launch (what ee_step_kernel does).
//
// Grids larger than the chip holds (12 x 1300 parts) give every workgroup a contiguous range of parts of its
// layer.  The data check (every value of the last stage against the recurrence run on the host) makes sure
// the flags really order the memory traffic: values live in a ring of two steps, so a slot is rewritten
// only after its three readers -- which lie in the writer's dependency cone -- are done.  Every wait is
// bounded: a workgroup that polls for more than ~2 s raises an error word and everything drains.
//
//   hipcc --offload-arch=gfx950 -O3 tools/flag_chain_probe.hip -o tools/_build/flag_chain_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
constexpr int kThreads = 256;
constexpr int kFlagStride = 16;  // uint32 per flag: 64-byte lines

__device__ __forceinline__ void spin_cycles(long long cycles) {
  if (cycles <= 0) return;
  const long long t0 = wall_clock64();  // 100 MHz constant clock
  const long long ticks = cycles / 24;  // ~2.4 GHz shader clock
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

__device__ __forceinline__ uint32_t stage_value(const uint32_t *prev, int p, int parts) {
  return prev[(p + parts - 1) % parts] + prev[p] + prev[(p + 1) % parts] + 1u;
}

// `pieces` 16-byte pieces from src to LDS by the waves first_wave .. 3 of the workgroup (the tail re-reads the last
// piece into the padding behind the slab)
__device__ __forceinline__ void stage_slab(const float4 *src, float4 *lds, int pieces, int first_wave = 0) {
  typedef __attribute__((address_space(3))) void lds_ptr_t;
  typedef const __attribute__((address_space(1))) void glb_ptr_t;
  const int wave = threadIdx.x / 64 - first_wave, lane = threadIdx.x & 63, nw = kThreads / 64 - first_wave;
  if (wave < 0) return;
  for (int p0 = 0; p0 < pieces; p0 += nw * 64) {
    const int i = p0 + wave * 64 + lane < pieces ? p0 + wave * 64 + lane : pieces - 1;
    __builtin_amdgcn_global_load_lds((glb_ptr_t *)(src + i), (lds_ptr_t *)(lds + p0 + wave * 64), 16, 0, 0);
  }
}

__global__ __launch_bounds__(kThreads) void stage_kernel(const uint32_t *__restrict__ prev, uint32_t *__restrict__ cur,
                                                         int parts, long long work, const float4 *slab, int pieces,
                                                         float *sink) {
  extern __shared__ float4 slab_lds[];
  const int p = blockIdx.x;
  if (pieces) {
    stage_slab(slab, slab_lds, pieces);
    __syncthreads();  // (vmcnt(0) + barrier: the slab is needed by the arithmetic)
  }
  spin_cycles(work);
  if (pieces && slab_lds[threadIdx.x % pieces].x == 12345.f) sink[0] = 1.f;  // (keeps the staging alive)
  if (threadIdx.x == 0) cur[p] = stage_value(prev, p, parts);
}

// One resident workgroup per part; flag[p * kFlagStride] = stages part p has completed (global stage count).
__global__ __launch_bounds__(kThreads) void owner_kernel(uint32_t *val, unsigned *flag, unsigned *error, int layers, int parts,
                                                         int steps, long long work, const float4 *slabs, int pieces,
                                                         float *sink) {
  extern __shared__ float4 slab_lds[];
  const int p = blockIdx.x;
  const int left = (p + parts - 1) % parts, right = (p + 1) % parts;
  __shared__ int bail;
  if (threadIdx.x == 0) bail = 0;
  if (pieces) stage_slab(slabs, slab_lds, pieces);  // stage 0's slab
  __syncthreads();
  const int ring = 2 * layers;
  const long long total = (long long)steps * layers;
  for (long long g = 0; g < total; g++) {
    if (threadIdx.x == 0 && g > 0) {
      long long spins = 0;
      for (int side = 0; side < 2 && !bail; side++) {
        const unsigned *f = flag + (size_t)(side ? right : left) * kFlagStride;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)g) {
          __builtin_amdgcn_s_sleep(1);
          if ((++spins & 1023) == 0 &&
              (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > (1ll << 24))) {
            __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bail = 1;
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // (also: this stage's slab has landed -- vmcnt(0) of every wave)
    if (bail) return;  // uniform
    spin_cycles(work);
    if (pieces && slab_lds[threadIdx.x % pieces].x == 12345.f) sink[0] = 1.f;
    __syncthreads();  // the arithmetic has read the slab: the next stage's may overwrite it, under publish + wait
    // (waves 1-3 move the slab; wave 0 publishes and polls: its vmcnt(0) waits must not sit behind a DMA of its own)
    if (pieces && g + 1 < total) stage_slab(slabs + (size_t)((g + 1) % layers) * pieces, slab_lds, pieces, 1);
    if (threadIdx.x == 0) {
      const uint32_t *prev = val + ((g + ring - 1) % ring) * parts;
      uint32_t *cur = val + (g % ring) * parts;
      cur[p] = stage_value(prev, p, parts);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(flag + (size_t)p * kFlagStride, (unsigned)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// val[2 * layers slots][parts]: slot of global stage g = g % (2 * layers); flag[(l * parts + p) * kFlagStride] =
// steps part p of layer l has completed.  central != 0: done[l * 32] counts the parts of layer l instead.
__global__ __launch_bounds__(kThreads) void persistent_kernel(uint32_t *val, unsigned *flag, unsigned *done, unsigned *error,
                                                              int layers, int parts, int steps, long long work,
                                                              int xcd_stride, int wg_per_layer, int central) {
  int b = blockIdx.x;
  if (xcd_stride > 1) {
    if (b % xcd_stride) return;  // not on the chosen XCD
    b /= xcd_stride;
  }
  const int l = b / wg_per_layer, slice = b % wg_per_layer;
  const int per = (parts + wg_per_layer - 1) / wg_per_layer;
  const int p_lo = slice * per, p_hi = p_lo + per < parts ? p_lo + per : parts;
  __shared__ int bail;
  if (threadIdx.x == 0) bail = 0;
  __syncthreads();
  const int ring = 2 * layers;
  for (int s = 0; s < steps; s++) {
    const long long stage = (long long)s * layers + l;
    const int pl = l == 0 ? layers - 1 : l - 1;           // producing layer
    const unsigned want = l == 0 ? (unsigned)s : (unsigned)(s + 1);  // steps it must have completed
    for (int p = p_lo; p < p_hi; p++) {
      if (threadIdx.x == 0 && stage > 0) {
        long long spins = 0;
        if (central) {
          const unsigned all = want * (unsigned)parts;
          while (__hip_atomic_load(&done[pl * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < all) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023) == 0 &&
                (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > (1ll << 24))) {
              __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bail = 1;
              break;
            }
          }
        } else {
          for (int d = -1; d <= 1 && !bail; d++) {
            const int q = (p + d + parts) % parts;
            const unsigned *f = flag + ((size_t)pl * parts + q) * kFlagStride;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
              __builtin_amdgcn_s_sleep(1);
              if ((++spins & 1023) == 0 &&
                  (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > (1ll << 24))) {
                __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bail = 1;
                break;
              }
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (bail) return;  // uniform
      spin_cycles(work);
      if (threadIdx.x == 0) {
        const uint32_t *prev = val + ((stage + ring - 1) % ring) * parts;  // stage 0 of step 0: the initial values in slot ring-1
        uint32_t *cur = val + (stage % ring) * parts;
        cur[p] = stage_value(prev, p, parts);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (central)
          __hip_atomic_fetch_add(&done[l * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          __hip_atomic_store(flag + ((size_t)l * parts + p) * kFlagStride, (unsigned)(s + 1), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e__ = (x);                                                         \
    if (e__ != hipSuccess) {                                                      \
      printf("%s: %s\n", #x, hipGetErrorString(e__));                             \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

std::vector<uint32_t> expected(int layers, int parts, int steps) {
  std::vector<uint32_t> a(parts), b(parts);
  for (int p = 0; p < parts; p++) a[p] = (uint32_t)p;
  for (long long st = 0; st < (long long)steps * layers; st++) {
    for (int p = 0; p < parts; p++) b[p] = a[(p + parts - 1) % parts] + a[p] + a[(p + 1) % parts] + 1u;
    a.swap(b);
  }
  return a;
}
}  // namespace

int main(int argc, char **argv) {
  const int layers = 12;
  const int steps = argc > 1 ? atoi(argv[1]) : 780;
  int failed = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int max_wg = prop.multiProcessorCount * 8;  // co-resident 256-thread workgroups
  printf("# %s, %d CUs; %d layers x %d steps; token work per stage, workgroups of %d threads\n", prop.name,
         prop.multiProcessorCount, layers, steps, kThreads);
  printf("# %-8s %6s %8s %10s %14s %12s\n", "mode", "parts", "wg/layer", "work(cyc)", "us/stage", "check");
  const bool owner_only = argc > 2 && atoi(argv[2]) != 0;  // r5: the part-owner table only
  const int slab_bytes = 17408;                             // ee_step_kernel<42, 17>: 1088 slots x 16 bytes
  float4 *slabs;
  float *sink;
  CHECK(hipMalloc(&slabs, (size_t)layers * slab_bytes));
  CHECK(hipMemset(slabs, 0, (size_t)layers * slab_bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(owner_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, slab_bytes + 4096));
  if (owner_only) {
    printf("# part-owner layout (one resident workgroup per part walks the %d layers); slab = %d bytes staged per stage\n", layers, slab_bytes);
    printf("# %-12s %6s %10s %8s %14s %12s\n", "mode", "parts", "work(cyc)", "slab", "us/stage", "check");
    for (int parts : {16, 42, 168, 650, 1300}) {
      for (long long work : {0ll, 8000ll}) {
        for (int pieces : {0, slab_bytes / 16}) {
          const int ring = 2 * layers;
          uint32_t *val;
          unsigned *flag, *error;
          CHECK(hipMalloc(&val, (size_t)ring * parts * 4));
          CHECK(hipMalloc(&flag, (size_t)parts * kFlagStride * 4));
          CHECK(hipMalloc(&error, 4));
          std::vector<uint32_t> init(parts);
          for (int p = 0; p < parts; p++) init[p] = (uint32_t)p;
          const std::vector<uint32_t> want = expected(layers, parts, steps);
          hipEvent_t e0, e1;
          CHECK(hipEventCreate(&e0));
          CHECK(hipEventCreate(&e1));
          for (int mode = 0; mode < 2; mode++) {  // 0: launch per stage, 1: part owners
            if (mode == 1 && parts > max_wg * 3 / 4) continue;
            CHECK(hipMemset(val, 0, (size_t)ring * parts * 4));
            CHECK(hipMemcpy(val + (size_t)(ring - 1) * parts, init.data(), parts * 4, hipMemcpyHostToDevice));
            CHECK(hipMemset(flag, 0, (size_t)parts * kFlagStride * 4));
            CHECK(hipMemset(error, 0, 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            if (mode == 0) {
              for (long long st = 0; st < (long long)steps * layers; st++)
                hipLaunchKernelGGL(stage_kernel, dim3(parts), dim3(kThreads), pieces ? pieces * 16 + 4096 : 0, 0, val + ((st + ring - 1) % ring) * parts,
                                   val + (st % ring) * parts, parts, work, slabs + (size_t)(st % layers) * pieces, pieces, sink);
            } else {
              hipLaunchKernelGGL(owner_kernel, dim3(parts), dim3(kThreads), pieces ? pieces * 16 + 3072 : 0, 0, val, flag, error, layers, parts, steps,
                                 work, slabs, pieces, sink);
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned err = 0;
            CHECK(hipMemcpy(&err, error, 4, hipMemcpyDeviceToHost));
            std::vector<uint32_t> got(parts);
            const long long last = (long long)steps * layers - 1;
            CHECK(hipMemcpy(got.data(), val + (last % ring) * parts, parts * 4, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int p = 0; p < parts; p++) bad += got[p] != want[p];
            failed += (bad != 0) || err;
            printf("  %-12s %6d %10lld %8d %14.3f %12s\n", mode == 0 ? "launch" : "owner", parts, work, pieces * 16,
                   ms * 1e3 / ((double)steps * layers), err ? "TIMED OUT" : (bad ? "WRONG DATA" : "ok"));
            fflush(stdout);
          }
          CHECK(hipFree(val));
          CHECK(hipFree(flag));
          CHECK(hipFree(error));
        }
      }
    }
    return failed != 0;
  }
  for (int parts : {16, 42, 168, 1300}) {
    for (long long work : {0ll, 8000ll}) {  // 8000 cycles ~ 3.3 us: a layer's own work at one frame
      const int ring = 2 * layers;
      uint32_t *val;
      unsigned *flag, *done, *error;
      CHECK(hipMalloc(&val, (size_t)ring * parts * 4));
      CHECK(hipMalloc(&flag, (size_t)layers * parts * kFlagStride * 4));
      CHECK(hipMalloc(&done, layers * 32 * 4));
      CHECK(hipMalloc(&error, 4));
      std::vector<uint32_t> init(parts);
      for (int p = 0; p < parts; p++) init[p] = (uint32_t)p;
      const std::vector<uint32_t> want = expected(layers, parts, steps);
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      const char *names[4] = {"launch", "p2p", "central", "p2p1x"};
      for (int mode = 0; mode < 4; mode++) {
        // persistent modes: every workgroup must be resident (they wait for each other)
        int wg_per_layer = parts;
        const int budget = (mode == 3 ? max_wg / 8 : max_wg) * 3 / 4;  // margin: never rely on the last slot
        while (layers * wg_per_layer > budget) wg_per_layer = (wg_per_layer + 1) / 2;
        if (mode == 3 && parts > 42) {
          printf("  %-8s %6d %8s %10lld %14s %12s\n", names[mode], parts, "-", work, "-", "(skipped)");
          continue;
        }
        CHECK(hipMemset(val, 0, (size_t)ring * parts * 4));
        CHECK(hipMemcpy(val + (size_t)(ring - 1) * parts, init.data(), parts * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(flag, 0, (size_t)layers * parts * kFlagStride * 4));
        CHECK(hipMemset(done, 0, layers * 32 * 4));
        CHECK(hipMemset(error, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        if (mode == 0) {
          for (long long st = 0; st < (long long)steps * layers; st++)
            hipLaunchKernelGGL(stage_kernel, dim3(parts), dim3(kThreads), 0, 0, val + ((st + ring - 1) % ring) * parts,
                               val + (st % ring) * parts, parts, work, (const float4 *)nullptr, 0, sink);
        } else {
          const int stride = mode == 3 ? 8 : 1;
          hipLaunchKernelGGL(persistent_kernel, dim3(layers * wg_per_layer * stride), dim3(kThreads), 0, 0, val, flag, done,
                             error, layers, parts, steps, work, stride, wg_per_layer, mode == 2 ? 1 : 0);
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err = 0;
        CHECK(hipMemcpy(&err, error, 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> got(parts);
        const long long last = (long long)steps * layers - 1;
        CHECK(hipMemcpy(got.data(), val + (last % ring) * parts, parts * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int p = 0; p < parts; p++) bad += got[p] != want[p];
        failed += (bad != 0) || err;
        printf("  %-8s %6d %8d %10lld %14.3f %12s\n", names[mode], parts, mode == 0 ? parts : wg_per_layer, work,
               ms * 1e3 / ((double)steps * layers), err ? "TIMED OUT" : (bad ? "WRONG DATA" : "ok"));
        fflush(stdout);
      }
      CHECK(hipFree(val));
      CHECK(hipFree(flag));
      CHECK(hipFree(done));
      CHECK(hipFree(error));
    }
  }
  return failed != 0;
}
