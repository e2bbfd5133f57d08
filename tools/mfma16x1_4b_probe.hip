// v_mfma_f32_16x16x1_4b_f32 (four independent 16 x 16 x 1 blocks per instruction): operand / result lane maps and
// chain exactness, for the encoder's lane-class GEMMs (csrc/entropy_mfma.hip): K = 17 needs 17 of these for FOUR
// classes instead of 4 x 5 v_mfma_f32_16x16x4_f32 (K padded to 20).  Assumed maps (checked here with random data
// against host fmaf chains): A lane L = A[block L >> 4][row L & 15], B lane L = B[block L >> 4][col L & 15],
// D register 4 b + r of lane L = D[block b][row 4 (L >> 4) + r][col L & 15].  Also times both forms (per-wave
// cycles for the same 4 classes x K = 17 x 6 tiles).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16x1_4b_probe tools/mfma16x1_4b_probe.hip && /tmp/mfma16x1_4b_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 17;

// a[4][16][K], b[4][K][16] -> d[4][16][16]
__global__ void probe(const float *a, const float *b, float *d) {
  const int l = threadIdx.x, r = l & 15, blk = l >> 4;
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  for (int k = 0; k < K; k++) {
    const float av = a[(blk * 16 + r) * K + k];
    const float bv = b[(blk * K + k) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(av, bv, acc, 0, 0, 0);
  }
  for (int bb = 0; bb < 4; bb++)
    for (int i = 0; i < 4; i++) d[(bb * 16 + 4 * blk + i) * 16 + r] = acc[4 * bb + i];
}

__global__ void time_4b(const float *a, const float *b, float *d, long long *cycles, int reps) {
  const int l = threadIdx.x;
  f32x16 acc[6];
  for (int t = 0; t < 6; t++)
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
  float av[3], bv[2];
  for (int t = 0; t < 3; t++) av[t] = a[l + 64 * t];
  for (int t = 0; t < 2; t++) bv[t] = b[l + 64 * t];
  const long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < reps; rep++)
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
      for (int mt = 0; mt < 3; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x1f32(av[mt], bv[nt], acc[mt * 2 + nt], 0, 0, 0);
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 6; t++)
    for (int i = 0; i < 16; i++) s += acc[t][i];
  d[blockIdx.x * 64 + l] = s;
  if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ void time_x4(const float *a, const float *b, float *d, long long *cycles, int reps) {
  const int l = threadIdx.x;
  f32x4 acc[6];
  for (int t = 0; t < 6; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av[3], bv[2];
  for (int t = 0; t < 3; t++) av[t] = a[l + 64 * t];
  for (int t = 0; t < 2; t++) bv[t] = b[l + 64 * t];
  const long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < reps; rep++)
#pragma unroll
    for (int cls = 0; cls < 4; cls++)
#pragma unroll
      for (int m = 0; m < 5; m++)
#pragma unroll
        for (int mt = 0; mt < 3; mt++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt * 2 + nt], 0, 0, 0);
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 6; t++)
    for (int i = 0; i < 4; i++) s += acc[t][i];
  d[blockIdx.x * 64 + l] = s;
  if (l == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  std::vector<float> a(4 * 16 * K), b(4 * K * 16), d(4 * 256);
  srand(7);
  for (auto &v : a) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : b) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *da, *db, *dd;
  long long *dc;
  hipMalloc(&da, a.size() * 4), hipMalloc(&db, b.size() * 4), hipMalloc(&dd, 1 << 20), hipMalloc(&dc, 8 * 1024);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 4; blk++)
    for (int r = 0; r < 16; r++)
      for (int c = 0; c < 16; c++) {
        float up = 0.f;
        for (int k = 0; k < K; k++) up = fmaf(a[(blk * 16 + r) * K + k], b[(blk * K + k) * 16 + c], up);
        bad += up != d[(blk * 16 + r) * 16 + c];
      }
  printf("v_mfma_f32_16x16x1_4b_f32, K = %d: %d of 1024 outputs differ from the k-ascending fmaf chain under the assumed lane maps\n", K, bad);
  const int reps = 200;
  for (int pass = 0; pass < 2; pass++) {
    long long c4b = 0, cx4 = 0;
    std::vector<long long> cyc(256);
    hipLaunchKernelGGL(time_4b, dim3(256), dim3(64), 0, 0, da, db, dd, dc, reps);
    hipMemcpy(cyc.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
    for (long long v : cyc) c4b += v;
    hipLaunchKernelGGL(time_x4, dim3(256), dim3(64), 0, 0, da, db, dd, dc, reps);
    hipMemcpy(cyc.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
    for (long long v : cyc) cx4 += v;
    printf("one wave per CU, 4 classes x 6 tiles: 16x16x1_4b %lld cycles per rep (102 MFMAs: %.1f each), 16x16x4 %lld (120 MFMAs: %.1f each)\n",
           c4b / 256 / reps, (double)c4b / 256 / reps / 102, cx4 / 256 / reps, (double)cx4 / 256 / reps / 120);
  }
  return 0;
}
