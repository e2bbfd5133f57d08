// Does v_mfma_f32_16x16x4_f32 accumulate its four products as ONE k-ascending fmaf chain (the numerics contract
// of the direct convolution kernel, which v_mfma_f32_32x32x2_f32 meets)?  Random operands, 64 chained MFMAs (K = 256)
// against fmaf chains on the host; prints the number of mismatching outputs for the ascending and the descending
// order inside an instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_chain_probe tools/mfma16_chain_probe.hip && /tmp/mfma16_chain_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 256;

// a[16][K], b[K][16] -> d[16][16]
__global__ void probe(const float *a, const float *b, float *d) {
  const int l = threadIdx.x, r = l % 16, kq = l / 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const float av = a[r * K + k0 + kq];        // A[row r][k0 + kq]
    const float bv = b[(k0 + kq) * 16 + r];     // B[k0 + kq][col r]
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
  }
  // D: lane l holds rows 4 (l / 16) + i, column l % 16
  for (int i = 0; i < 4; i++) d[(4 * kq + i) * 16 + r] = acc[i];
}

int main() {
  std::vector<float> a(16 * K), b(K * 16), d(256);
  srand(7);
  for (auto &v : a) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : b) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *da, *db, *dd;
  hipMalloc(&da, a.size() * 4), hipMalloc(&db, b.size() * 4), hipMalloc(&dd, 256 * 4);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost);
  int bad_up = 0, bad_down = 0;
  double worst = 0;
  for (int r = 0; r < 16; r++)
    for (int c = 0; c < 16; c++) {
      float up = 0.f, down = 0.f;
      for (int k0 = 0; k0 < K; k0 += 4) {
        for (int q = 0; q < 4; q++) up = fmaf(a[r * K + k0 + q], b[(k0 + q) * 16 + c], up);
        for (int q = 3; q >= 0; q--) down = fmaf(a[r * K + k0 + q], b[(k0 + q) * 16 + c], down);
      }
      bad_up += up != d[r * 16 + c];
      bad_down += down != d[r * 16 + c];
      worst = fmax(worst, fabs((double)up - d[r * 16 + c]));
    }
  printf("v_mfma_f32_16x16x4_f32, K = %d: %d of 256 outputs differ from the k-ascending fmaf chain (max |diff| %.3g), %d from the descending one\n",
         K, bad_up, worst, bad_down);
  return 0;
}
