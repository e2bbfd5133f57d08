// What HBM delivers for the access pattern of the 1x1 / GDN layers: a workgroup reads a tile of C channels x R rows
// x W pixels of an NCHW tensor (W * 4 bytes contiguous, then a jump of a row or a channel), reads the same tile
// of a second tensor and writes their sum to a third -- no arithmetic worth mentioning, 16-byte accesses, enough
// workgroups to fill the chip.  Prints GB/s per tile width; W = 64 is the convolution kernels' tile.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_pattern_probe tools/hbm_pattern_probe.hip && /tmp/hbm_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W, int R, bool TWO>
__global__ __launch_bounds__(256) void tile_sum(const float *__restrict__ a, const float *__restrict__ b,
                                                float *__restrict__ o, int C, int h, int w, int tiles_r, int tiles_c) {
  int t = blockIdx.x;
  const int tr = t % tiles_r;
  t /= tiles_r;
  const int tc = t % tiles_c;
  const int img = t / tiles_c;
  constexpr int LPS = W / 4;            // lanes per segment
  constexpr int SEGS = 256 / LPS;       // segments in flight per iteration
  const int seg0 = threadIdx.x / LPS, l = threadIdx.x % LPS;
  const size_t base = (size_t)img * C * h * w + (size_t)(tr * R) * w + tc * W + l * 4;
  for (int s = seg0; s < C * R; s += SEGS) {
    const int ch = s / R, row = s % R;
    const size_t off = base + (size_t)ch * h * w + (size_t)row * w;
    float4 x = *reinterpret_cast<const float4 *>(a + off);
    if (TWO) {
      const float4 y = *reinterpret_cast<const float4 *>(b + off);
      x.x += y.x, x.y += y.y, x.z += y.z, x.w += y.w;
    }
    *reinterpret_cast<float4 *>(o + off) = x;
  }
}

template <int W, int R, bool TWO>
void run(const float *a, const float *b, float *o, int n, int C, int h, int w) {
  const int tiles_r = h / R, tiles_c = w / W;
  const int grid = n * tiles_r * tiles_c;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((tile_sum<W, R, TWO>), dim3(grid), dim3(256), 0, 0, a, b, o, C, h, w, tiles_r, tiles_c);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((tile_sum<W, R, TWO>), dim3(grid), dim3(256), 0, 0, a, b, o, C, h, w, tiles_r, tiles_c);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = (double)n * C * h * w * 4 * (TWO ? 3 : 2);
  printf("tile %3d px x %d rows x %d ch, %s: %.3f ms  %.0f GB/s\n", W, R, C, TWO ? "a + b -> o" : "a -> o     ", ms, bytes / ms * 1e-6);
}

int main() {
  const int n = 128, C = 192, h = 32, w = 1024;   // the codec's quarter-scale batch: 3.2 GB per tensor
  const size_t elems = (size_t)n * C * h * w;
  float *a, *b, *o;
  if (hipMalloc(&a, elems * 4) != hipSuccess || hipMalloc(&b, elems * 4) != hipSuccess || hipMalloc(&o, elems * 4) != hipSuccess) {
    printf("hipMalloc failed\n");
    return 1;
  }
  hipMemset(a, 0, elems * 4), hipMemset(b, 0, elems * 4);
  run<64, 2, true>(a, b, o, n, C, h, w);
  run<128, 1, true>(a, b, o, n, C, h, w);
  run<256, 1, true>(a, b, o, n, C, h, w);
  run<64, 2, false>(a, b, o, n, C, h, w);
  run<64, 4, false>(a, b, o, n, C, h, w);
  run<128, 1, false>(a, b, o, n, C, h, w);
  run<256, 1, false>(a, b, o, n, C, h, w);
  // the whole tensor as one run (tiles of 1024 px = whole rows, 32 rows: contiguous per channel)
  run<256, 32, false>(a, b, o, n, C, h, w);
  hipFree(a), hipFree(b), hipFree(o);
  return 0;
}
