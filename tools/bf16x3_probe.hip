// Split-bf16 tile convolution, PROBE ONLY (VERDICT r4 "next round" item 5): never the default, never the headline.
// The fp32 matrix instruction runs at 1/16 of the bf16 rate; a * b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi with
// a_hi = bf16(a), a_lo = bf16(a - a_hi) on v_mfma_f32_32x32x16_bf16 (fp32 accumulation) is three instructions of
// 32 cycles per 16 k against eight of 64: 5.3 x the fp32 matrix peak at ~2^-16 relative error per product.
//   (1) error: the GEMM of a 3x3 192 -> 192 layer (M = 192 couts, K = 1728, N = pixels; the convolution IS this
//       product over im2col columns) against float64, relative to the output scale sqrt(sum w^2 x^2), at the four
//       scales of tests/test_gpu_wino42.py::test_wino42_relative_error_at_other_scales + unit scale, next to the
//       k-ascending fp32 fmaf chain (the direct kernel's numerics);
//   (2) rate: a matrix-bound loop, operands re-read from LDS every step (2 x 2 register blocking), split-bf16
//       against v_mfma_f32_32x32x2_f32: the ceiling a tuned kernel could approach, not a kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/bf16x3_probe tools/bf16x3_probe.hip && tools/_build/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline uint16_t to_bf16(float x) {  // round to nearest even (finite inputs)
  uint32_t u;
  memcpy(&u, &x, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float from_bf16(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// C[M][N] = A[M][K] x B[K][N]; one wave per 32 x 32 tile; mode 0: split-bf16 (3 MFMAs per 16 k), 1: bf16 only (1)
__global__ void gemm_split(const float *A, const float *B, float *C, int M, int N, int K, int mode) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; j++) {
      const int k = k0 + 8 * h + j;
      const float a = A[(size_t)(m0 + r) * K + k], b = B[(size_t)k * N + n0 + r];
      const uint16_t a1 = to_bf16(a), b1 = to_bf16(b);
      ah[j] = (short)a1, bh[j] = (short)b1;
      al[j] = (short)to_bf16(a - from_bf16(a1)), bl[j] = (short)to_bf16(b - from_bf16(b1));
    }
    if (mode == 0) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; i++) C[(size_t)(m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n0 + r] = acc[i];
}

// matrix-bound loops: 2 x 2 tiles of 32 x 32 per wave, fragments re-read from LDS every k step
__global__ __launch_bounds__(256) void rate_bf16x3(const float *seed, float *sink, int steps) {
  __shared__ __attribute__((aligned(16))) bf16x8 frag[8][64 + 1];  // a_hi[2] a_lo[2] b_hi[2] b_lo[2] (padded rows)
  const int l = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8 * 65; i += 256) {
    bf16x8 v;
    for (int j = 0; j < 8; j++) v[j] = (short)to_bf16(seed[(i * 8 + j) & 1023]);
    (&frag[0][0])[i] = v;
  }
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; t++)
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
  for (int s = 0; s < steps; s++) {
    bf16x8 f[8];
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = frag[k][(l + s) & 63];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 + mt], f[4 + nt], acc[mt * 2 + nt], 0, 0, 0);
        acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[mt], f[6 + nt], acc[mt * 2 + nt], 0, 0, 0);
        acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[mt], f[4 + nt], acc[mt * 2 + nt], 0, 0, 0);
      }
  }
  float sum = 0;
  for (int t = 0; t < 4; t++)
    for (int i = 0; i < 16; i++) sum += acc[t][i];
  sink[blockIdx.x * 256 + threadIdx.x] = sum;
}

__global__ __launch_bounds__(256) void rate_f32(const float *seed, float *sink, int steps) {
  __shared__ float frag[4][16][64 + 1];  // a[2], b[2]: 16 k-pairs of one float per lane
  const int l = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4 * 16 * 65; i += 256) (&frag[0][0][0])[i] = seed[i & 1023];
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; t++)
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
  for (int s = 0; s < steps; s++) {  // one step = 16 k = 8 instructions per tile
#pragma unroll
    for (int kp = 0; kp < 8; kp++) {
      float a[2], b[2];
      a[0] = frag[0][kp][(l + s) & 63], a[1] = frag[1][kp][(l + s) & 63];
      b[0] = frag[2][kp][(l + s) & 63], b[1] = frag[3][kp][(l + s) & 63];
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt * 2 + nt], 0, 0, 0);
    }
  }
  float sum = 0;
  for (int t = 0; t < 4; t++)
    for (int i = 0; i < 16; i++) sum += acc[t][i];
  sink[blockIdx.x * 256 + threadIdx.x] = sum;
}

#define CHECK(x)                                                    \
  do {                                                              \
    hipError_t e__ = (x);                                           \
    if (e__ != hipSuccess) {                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e__));               \
      exit(2);                                                      \
    }                                                               \
  } while (0)

int main() {
  const int M = 192, K = 1728, N = 1024;
  std::mt19937 rng(21);
  std::normal_distribution<float> nd(0.f, 1.f);
  float *dA, *dB, *dC;
  CHECK(hipMalloc(&dA, (size_t)M * K * 4));
  CHECK(hipMalloc(&dB, (size_t)K * N * 4));
  CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  printf("# split-bf16 GEMM of a 3x3 192 -> 192 layer (M %d, K %d, N %d) against float64: max |err| / sqrt(sum w^2 x^2)\n", M, K, N);
  printf("# %-22s %14s %14s %14s\n", "x scale, w scale", "split-bf16 (3)", "bf16 only (1)", "fp32 fmaf chain");
  const double scales[5][2] = {{1, 1}, {1e3, 1}, {1e-3, 1}, {1, 1e3}, {30, 30}};
  for (int sc = 0; sc < 5; sc++) {
    std::vector<float> A((size_t)M * K), B((size_t)K * N), C((size_t)M * N);
    for (int m = 0; m < M; m++)
      for (int k = 0; k < K; k++) {
        float v = nd(rng) * (float)(scales[sc][1] / sqrt((double)K));
        if (m % 4 == 0) v *= 100.f;          // mixed-scale weights, as in the Winograd tests
        if ((k / 9) % 3 == 0) v *= 0.01f;
        A[(size_t)m * K + k] = v;
      }
    for (auto &v : B) v = nd(rng) * (float)scales[sc][0];
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    double worst[3] = {0, 0, 0};
    for (int mode = 0; mode < 2; mode++) {
      hipLaunchKernelGGL(gemm_split, dim3(N / 32, M / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K, mode);
      CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
      for (int m = 0; m < M; m += 3)
        for (int n = 0; n < N; n += 7) {
          double ref = 0, s2 = 0;
          float chain = 0.f;
          for (int k = 0; k < K; k++) {
            const double a = A[(size_t)m * K + k], b = B[(size_t)k * N + n];
            ref += a * b;
            s2 += a * a * b * b;
            if (mode == 0) chain = fmaf(A[(size_t)m * K + k], B[(size_t)k * N + n], chain);
          }
          worst[mode] = fmax(worst[mode], fabs(C[(size_t)m * N + n] - ref) / sqrt(s2));
          if (mode == 0) worst[2] = fmax(worst[2], fabs((double)chain - ref) / sqrt(s2));
        }
    }
    printf("  %-8g %-13g %14.3g %14.3g %14.3g\n", scales[sc][0], scales[sc][1], worst[0], worst[1], worst[2]);
  }
  // rates
  float *seed, *sink;
  std::vector<float> hs(1024);
  for (auto &v : hs) v = nd(rng);
  CHECK(hipMalloc(&seed, 4096));
  CHECK(hipMalloc(&sink, 2048 * 256 * 4));
  CHECK(hipMemcpy(seed, hs.data(), 4096, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int steps = 4000, blocks = 256 * 2;  // 2 workgroups of 4 waves per CU
  printf("# matrix-bound loops, %d workgroups x 4 waves, 2 x 2 tiles of 32 x 32 per wave, operands from LDS every step\n", blocks);
  for (int pass = 0; pass < 2; pass++) {
    float ms3 = 0, ms1 = 0;
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_bf16x3, dim3(blocks), dim3(256), 0, 0, seed, sink, steps);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms3, e0, e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_f32, dim3(blocks), dim3(256), 0, 0, seed, sink, steps / 4);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms1, e0, e1));
    const double flop3 = 2.0 * 64 * 64 * 16 * (double)steps * 4 * blocks, flop1 = 2.0 * 64 * 64 * 16 * (double)(steps / 4) * 4 * blocks;
    printf("  split-bf16: %.1f TFLOP/s of fp32-equivalent products (%.1f executed bf16);  fp32 MFMA: %.1f TFLOP/s;  ratio %.2f\n",
           flop3 / ms3 / 1e9, 3 * flop3 / ms3 / 1e9, flop1 / ms1 / 1e9, (flop3 / ms3) / (flop1 / ms1));
  }
  return 0;
}
