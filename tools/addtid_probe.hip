// Where does ds_write_addtid_b32 write?  (address = M0[15:0] + offset + 4 * lane, or 4 * thread of the workgroup?)
//   hipcc --offload-arch=gfx950 -O3 tools/addtid_probe.hip -o tools/_build/addtid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 36864;  // floats of LDS (144 KB)
template <int OFF>
__device__ __forceinline__ void put(float v, unsigned m0v) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2\n\ts_waitcnt lgkmcnt(0)" ::"v"(v), "s"(m0v), "n"(OFF) : "memory");
}
__global__ __launch_bounds__(512) void k(float *out, int test) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < N; i += 512) lds[i] = -1.f;
  __syncthreads();
  const float v = (float)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (test == 0) put<0>(v, 0u);
  if (test == 1) put<512>(v, 1024u);
  if (test == 2) put<0>(v, 0x10000u + 16u);
  if (test == 3) put<60000>(v, 60000u);
  if (test == 4) put<256>(v, (unsigned)wave * 8192u);
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 512) out[i] = lds[i];
  if (threadIdx.x == 0) out[N] = (float)(unsigned)reinterpret_cast<uintptr_t>(lds);
}
int main() {
  float *d;
  hipMalloc(&d, (N + 1) * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, N * 4);
  std::vector<float> h(N + 1);
  for (int t = 0; t < 5; t++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(512), N * 4, 0, d, t);
    hipMemcpy(h.data(), d, (N + 1) * 4, hipMemcpyDeviceToHost);
    printf("test %d (lds base %g): ", t, h[N]);
    int shown = 0;
    for (int i = 0; i < N && shown < 24; i++)
      if (h[i] >= 0.f && ((int)h[i] % 64 == 0 || (int)h[i] % 64 == 63)) { printf("[byte %d]=%g ", i * 4, h[i]); shown++; }
    int cnt = 0;
    for (int i = 0; i < N; i++) cnt += h[i] >= 0.f;
    printf(" (%d floats written)\n", cnt);
  }
  return 0;
}
