// What does a dependency between two "layers" of the entropy decoder cost when it is NOT a kernel
// launch?  (SURVEY 8f-1 / DESIGN 5: the persistent step kernel was priced from the guide's GRID
// barrier; the dependency is neighbour-to-neighbour, i.e. point-to-point flags, which this measures.)
//
// A step of the decoder is LAYERS dependent stages; stage l of step s may start when stage l-1 of
// step s is complete (stage 0: when the last stage of step s-1 is).  Three ways to run STEPS x LAYERS
// stages of PARTS workgroups each, all doing the same token work (read two values the previous stage
// wrote -- its own part's and a neighbour's -- add, write; optionally spin `work` cycles):
//
//   launch    one kernel launch per stage, in one stream (what the engine does today)
//   flags     ONE persistent kernel of LAYERS x PARTS workgroups; a stage's workgroups bump a device
//             counter (release, agent scope) when done and the next stage's poll it (acquire)
//   flags1x   the same with every workgroup on ONE XCD (workgroups are dealt to the 8 XCDs round-robin
//             by block index: the grid is 8 x larger and only blocks with index % 8 == 0 take part),
//             so that no flag or datum crosses an L2 boundary
//
// The data check (every value of the last stage against the closed form computed on the host) makes
// sure the flags really order the stages' memory traffic.  Every wait is bounded: a workgroup that
// polls for more than ~2 s raises an error word and everything drains.
//
//   hipcc --offload-arch=gfx950 -O3 tools/flag_chain_probe.hip -o tools/_build/flag_chain_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
constexpr int kThreads = 256;

__device__ __forceinline__ void spin_cycles(long long cycles) {
  if (cycles <= 0) return;
  const long long t0 = wall_clock64();  // 100 MHz constant clock
  const long long ticks = cycles / 24;  // ~2.4 GHz shader clock
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

// value of (layer l, part p) of step s from the previous stage's values
__device__ __forceinline__ uint32_t stage_value(const uint32_t *prev, int p, int parts) {
  return prev[p] + prev[(p + 1) % parts] + 1u;
}

__global__ __launch_bounds__(kThreads) void stage_kernel(const uint32_t *__restrict__ prev, uint32_t *__restrict__ cur,
                                                         int parts, long long work) {
  const int p = blockIdx.x;
  spin_cycles(work);
  if (threadIdx.x == 0) cur[p] = stage_value(prev, p, parts);
}

// buffers: val[(LAYERS + 1) ring slots][parts]; slot of (step s, layer l) = global stage index % ring
__global__ __launch_bounds__(kThreads) void persistent_kernel(uint32_t *val, unsigned *done, unsigned *error, int layers,
                                                              int parts, int steps, long long work, int xcd_stride) {
  int b = blockIdx.x;
  if (xcd_stride > 1) {
    if (b % xcd_stride) return;  // not on the chosen XCD
    b /= xcd_stride;
  }
  const int l = b / parts, p = b % parts;
  __shared__ int bail;
  if (threadIdx.x == 0) bail = 0;
  __syncthreads();
  const int ring = layers + 1;
  for (int s = 0; s < steps; s++) {
    const long long stage = (long long)s * layers + l;  // global index of this stage; stage -1 = the initial values
    if (threadIdx.x == 0) {
      if (stage > 0) {
        // the previous stage is complete when its counter has seen all its parts
        const long long prev = stage - 1;
        const int pl = (int)(prev % layers);
        const unsigned want = (unsigned)((prev / layers + 1) * parts);
        long long spins = 0;
        while (__hip_atomic_load(&done[pl * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
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
    __syncthreads();
    if (bail) return;  // uniform
    spin_cycles(work);
    if (threadIdx.x == 0) {
      const uint32_t *prev = val + ((stage + ring - 1) % ring) * parts;  // slot of stage - 1 (stage 0: the initial values, slot ring-1)
      uint32_t *cur = val + (stage % ring) * parts;
      // (the acquire above makes the previous stage's stores visible; these are plain accesses)
      cur[p] = stage_value(prev, p, parts);
      __hip_atomic_fetch_add(&done[l * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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

// host closed form: run the recurrence
std::vector<uint32_t> expected(int layers, int parts, int steps) {
  std::vector<uint32_t> a(parts), b(parts);
  for (int p = 0; p < parts; p++) a[p] = (uint32_t)p;
  for (long long st = 0; st < (long long)steps * layers; st++) {
    for (int p = 0; p < parts; p++) b[p] = a[p] + a[(p + 1) % parts] + 1u;
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
  printf("# %s, %d CUs; %d layers x %d steps; token work per stage, workgroups of %d threads\n", prop.name,
         prop.multiProcessorCount, layers, steps, kThreads);
  printf("# %-8s %6s %10s %14s %12s\n", "mode", "parts", "work(cyc)", "us/stage", "check");
  for (int parts : {16, 42, 168}) {
    for (long long work : {0ll, 8000ll}) {  // 8000 cycles ~ 3.3 us: a layer's own work at one frame
      const int ring = layers + 1;
      uint32_t *val;
      unsigned *done, *error;
      CHECK(hipMalloc(&val, (size_t)ring * parts * 4));
      CHECK(hipMalloc(&done, layers * 32 * 4));
      CHECK(hipMalloc(&error, 4));
      std::vector<uint32_t> init(parts);
      for (int p = 0; p < parts; p++) init[p] = (uint32_t)p;
      const std::vector<uint32_t> want = expected(layers, parts, steps);
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      for (int mode = 0; mode < 3; mode++) {
        if (mode == 2 && layers * parts > 32 * 8) {
          // one XCD holds 32 CUs x 8 workgroups of this size: the large grid does not fit one XCD
          printf("  %-8s %6d %10lld %14s %12s\n", "flags1x", parts, work, "-", "(does not fit one XCD)");
          continue;
        }
        CHECK(hipMemset(val, 0, (size_t)ring * parts * 4));
        CHECK(hipMemcpy(val + (size_t)(ring - 1) * parts, init.data(), parts * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(done, 0, layers * 32 * 4));
        CHECK(hipMemset(error, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        if (mode == 0) {
          for (long long st = 0; st < (long long)steps * layers; st++)
            hipLaunchKernelGGL(stage_kernel, dim3(parts), dim3(kThreads), 0, 0, val + ((st + ring - 1) % ring) * parts,
                               val + (st % ring) * parts, parts, work);
        } else {
          const int stride = mode == 2 ? 8 : 1;
          hipLaunchKernelGGL(persistent_kernel, dim3(layers * parts * stride), dim3(kThreads), 0, 0, val, done, error,
                             layers, parts, steps, work, stride);
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
        printf("  %-8s %6d %10lld %14.3f %12s\n", mode == 0 ? "launch" : (mode == 1 ? "flags" : "flags1x"), parts, work,
               ms * 1e3 / ((double)steps * layers), err ? "TIMED OUT" : (bad ? "WRONG DATA" : "ok"));
        fflush(stdout);
      }
      CHECK(hipFree(val));
      CHECK(hipFree(done));
      CHECK(hipFree(error));
    }
  }
  return failed != 0;
}
