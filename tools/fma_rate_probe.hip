// Issue rate of plain vs packed fp32 FMAs on gfx950, by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/fma_rate_probe.hip -o tools/_build/fma_rate_probe
// Every lane runs 24 independent accumulator chains (the shape of a band-kernel round: 8 positions x 3
// outputs) for ITERS rounds; plain = 24 v_fma_f32 per round, packed = 12 v_pk_fma_f32 per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PACKED>
__global__ __launch_bounds__(256) void fma_kernel(float *out, int iters, float w0, float w1, float w2) {
  float acc[24];
  for (int i = 0; i < 24; i++) acc[i] = threadIdx.x * 1e-3f + i;
  float x[8];
  for (int p = 0; p < 8; p++) x[p] = 1.0f + 1e-6f * (threadIdx.x + p);
  for (int it = 0; it < iters; it++) {
    if (PACKED) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f2 xx = {x[2 * q], x[2 * q + 1]};
#pragma unroll
        for (int o = 0; o < 3; o++) {
          f2 a = {acc[(2 * q) * 3 + o], acc[(2 * q + 1) * 3 + o]};
          const float w = o == 0 ? w0 : (o == 1 ? w1 : w2);
          f2 ww = {w, w};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(xx), "v"(ww));
          acc[(2 * q) * 3 + o] = a.x;
          acc[(2 * q + 1) * 3 + o] = a.y;
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; p++)
#pragma unroll
        for (int o = 0; o < 3; o++) {
          const float w = o == 0 ? w0 : (o == 1 ? w1 : w2);
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[p * 3 + o]) : "v"(x[p]), "v"(w));
        }
    }
  }
  float s = 0;
  for (int i = 0; i < 24; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out;
  hipMalloc(&out, 256 * 256 * 8 * 4 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  printf("# waves/SIMD   plain TFLOP/s   packed TFLOP/s   (24 chains per lane, %d rounds)\n", iters);
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int blocks = 256 * wps;  // 256-thread blocks: one wave per SIMD each
    double tf[2];
    for (int m = 0; m < 2; m++) {
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        if (m == 0)
          hipLaunchKernelGGL(fma_kernel<false>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 1.001f, 0.9995f);
        else
          hipLaunchKernelGGL(fma_kernel<true>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 1.001f, 0.9995f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      tf[m] = 2.0 * 24 * 256.0 * blocks * iters / (ms * 1e-3) / 1e12;
    }
    printf("  %d   %10.1f   %10.1f\n", wps, tf[0], tf[1]);
  }
  return 0;
}
