// What the fp32 matrix pipes sustain when EVERY SIMD of the chip runs them back to back (the clock under that load
// is not the 2.4 GHz the 157.3 TFLOP/s peak is quoted at): independent v_mfma_f32_32x32x2_f32 / 16x16x4_f32 /
// 16x16x1_4b_f32 chains, 1 .. 4 independent chains per wave, 1 .. 4 waves per SIMD, launches of ~17 (or ~170) ms.  Reports TFLOP/s by HIP events and the
// shader clock from the kernel's own counters (s_memtime cycles per 100 MHz wall tick).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_sustained_probe tools/mfma_sustained_probe.hip && /tmp/mfma_sustained_probe [random|zeros [long]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FORM, int NACC>
__global__ __launch_bounds__(256) void burn(const float *in, float *out, long long *clk, int reps) {
  const int l = threadIdx.x;
  const float a = in[l], b = in[l + 256];
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  float s = 0.f;
  if (FORM == 0) {  // 32 x 32 x 2: 16 passes, 4096 flop
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; t++)
      for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    for (int r = 0; r < reps; r++)
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    for (int t = 0; t < NACC; t++)
      for (int i = 0; i < 16; i++) s += acc[t][i];
  } else if (FORM == 1) {  // 16 x 16 x 4: 8 passes, 2048 flop
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; r++)
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    for (int t = 0; t < NACC; t++)
      for (int i = 0; i < 4; i++) s += acc[t][i];
  } else {  // 16 x 16 x 1, four blocks: 8 passes, 2048 flop
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; t++)
      for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    for (int r = 0; r < reps; r++)
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[t], 0, 0, 0);
    for (int t = 0; t < NACC; t++)
      for (int i = 0; i < 16; i++) s += acc[t][i];
  }
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[blockIdx.x * 256 + l] = s;
  if (l == 0) clk[2 * blockIdx.x] = c1 - c0, clk[2 * blockIdx.x + 1] = w1 - w0;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
  const bool random = argc > 1 && argv[1][0] == 'r';
  const bool full = argc > 2;  // launches of ~170 ms instead of ~17  // operands: zeros, or ("random") values in +-1 / 64
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int wall_khz = 0;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("%s: %d CUs, clockRate %d kHz, wall clock %d kHz\n", prop.name, cus, prop.clockRate, wall_khz);
  float *in, *out;
  long long *clk;
  const int max_blocks = cus * 4;
  CHECK(hipMalloc(&in, 512 * 4));
  CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * 4));
  CHECK(hipMalloc(&clk, (size_t)max_blocks * 16));
  std::vector<float> h(512, 0.f);
  if (random) {
    unsigned v = 12345u;
    for (float &x : h) v = v * 1664525u + 1013904223u, x = ((int)(v >> 8) % 2001 - 1000) / 64000.f;
  }
  printf("operands: %s\n", random ? "random" : "zeros");
  CHECK(hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const char *names[3] = {"32x32x2", "16x16x4", "16x16x1_4b"};
  const double flop_per_mfma[3] = {4096.0, 2048.0, 2048.0};
  const int unroll[3] = {4, 8, 8};
  typedef void (*kernel_t)(const float *, float *, long long *, int);
  // independent accumulator chains per wave: a chain's next instruction needs the previous one's result
  const int naccs[4] = {1, 2, 3, 4};
  const kernel_t kernels[3][4] = {{burn<0, 1>, burn<0, 2>, burn<0, 3>, burn<0, 4>},
                                  {burn<1, 1>, burn<1, 2>, burn<1, 3>, burn<1, 4>},
                                  {burn<2, 1>, burn<2, 2>, burn<2, 3>, burn<2, 4>}};
  for (int form = 0; form < 3; form++)
    for (int ai = 0; ai < 4; ai++)
      for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD = workgroups (4 waves) per CU
        if (naccs[ai] != 4 && wps == 4) continue;
        const int nacc = naccs[ai];
        const int reps = (full ? 400000 : 40000) * 4 / nacc / wps;
        const int blocks = cus * wps;
        float ms = 0;
        for (int pass = 0; pass < 2; pass++) {
          CHECK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(kernels[form][ai], dim3(blocks), dim3(256), 0, 0, in, out, clk, reps);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<long long> c(2 * blocks);
        CHECK(hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0;
        for (int i = 0; i < blocks; i++) cyc += (double)c[2 * i], wall += (double)c[2 * i + 1];
        const double tflops = (double)blocks * 4 * reps * unroll[form] * nacc * flop_per_mfma[form] / (ms * 1e-3) / 1e12;
        printf("%-11s %d chain(s) x %d wave(s)/SIMD %8.2f ms: %6.1f TFLOP/s = %.3f of 157.3; %.0f MHz\n", names[form], nacc,
               wps, ms, tflops, tflops / 157.3, cyc / wall * wall_khz / 1e3);
        fflush(stdout);
      }
  return 0;
}
