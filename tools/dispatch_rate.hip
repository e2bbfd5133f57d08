// How fast does the chip start workgroups?  Empty kernels with the launch shape of
// the tile convolution (256 threads, N KB of dynamic LDS), timed with HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) empty_kernel(float *out, int touch) {
  extern __shared__ float lds[];
  if (touch && threadIdx.x == 0) lds[0] = 1.f;
  if (out && blockIdx.x == 0x7fffffff) out[0] = lds[0];
}

__global__ void __launch_bounds__(256) zero_kernel(float *out, int per_wg) {
  extern __shared__ float lds[];
  float *p = out + (size_t)blockIdx.x * per_wg;
  for (int e = threadIdx.x; e < per_wg; e += 256) p[e] = 0.f;
}

int main() {
  float *buf;
  hipMalloc(&buf, (size_t)16384 * 24576 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int lds_kb[] = {0, 16, 64, 80, 128};
  for (int li = 0; li < 5; ++li) {
    size_t smem = (size_t)lds_kb[li] * 1024;
    hipFuncSetAttribute((const void *)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void *)zero_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int grid = 16384; grid <= 65536; grid *= 4) {
      float ms;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), smem, 0, buf, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("empty  lds %3d KB grid %6d: %.3f ms/launch  %.3f us/WG\n", lds_kb[li], grid, ms / 10, ms / 10 * 1e3 / grid);
    }
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(zero_kernel, dim3(16384), dim3(256), smem, 0, buf, 24576);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("zero96KB lds %3d KB grid 16384: %.3f ms/launch  %.3f us/WG\n", lds_kb[li], ms / 10, ms / 10 * 1e3 / 16384);
  }
  return 0;
}
