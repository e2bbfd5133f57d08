// VERDICT r4 item 3, measured: what a PRE-TRANSFORM pass would cost the F(4x2,3x3) kernel of a 192 -> 768 layer.
// Today every 64-cout workgroup (12 per tile block for 768 couts) fetches the 10 x 66 patch of 4 channels per chunk
// (10.6 KB by LDS-DMA) and transforms it in its loop; a pre-transform pass would write V = Bt6 d B4 once
// (24 values per 4 x 2 tile and channel: 3 x the input bytes) and the GEMM kernel would stream V (24.6 KB per chunk)
// with no transform.  This probe times, at the layer's shape (16 tiles x 192 channels x 34 x 1026, one frame):
//   (1) the pre-transform pass itself (reads the input once, writes V in the GEMM kernel's stage layout);
//   (2) a DMA-only skeleton of the GEMM kernel's operand traffic, one workgroup per CU (144 KB of LDS) as the real
//       kernel: 48 chunks, double buffered, (a) the patch form: 6 dwords per thread and chunk, (b) the V form: 3
//       16-byte pieces per thread and chunk -- 12 cout blocks per tile block, launch order cout-block fastest and
//       XCD-grouped (the cout blocks of a tile block on one XCD, so that its L2 serves eleven of the twelve reads).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wino42_pretransform_probe tools/wino42_pretransform_probe.hip && /tmp/wino42_pretransform_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

constexpr int TN = 16, C = 192, H = 34, W = 1026;      // input of the layer (one frame: 16 tiles)
constexpr int HO = H - 2, WO = W - 2;                  // 32 x 1024 outputs
constexpr int RB = HO / 8, CB = WO / 64;               // 4 x 16 tile blocks of 8 x 64 outputs per image tile
constexpr int NBLK = TN * RB * CB;                     // 1024
constexpr int KC = 4, NCHUNK = C / KC;                 // 48 chunks of 4 channels
constexpr int PR = 10, PC = 66, NXI = 24, NT = 64;
constexpr int VSZ = NXI * KC * NT;                     // 6144 floats per (block, chunk)

// (1) V[block][chunk][xi = 4 i + j][ci][tile = 32 ty + tx]: tile (ty, tx) = rows 4 ty .. 4 ty + 5, columns 2 tx .. 2 tx + 3
__global__ __launch_bounds__(256) void pretransform(const float *__restrict__ x, float *__restrict__ v) {
  __shared__ float patch[KC][PR][PC + 2];
  const int blk = blockIdx.x;
  const int cb = blk % CB, rb = (blk / CB) % RB, t = blk / (CB * RB);
  const int tid = threadIdx.x, ci = tid >> 6, tile = tid & 63, ty = tile >> 5, tx = tile & 31;
  const float *xt = x + (size_t)t * C * H * W + (size_t)(8 * rb) * W + 64 * cb;
  float *vb = v + (size_t)blk * NCHUNK * VSZ;
  for (int ch = 0; ch < NCHUNK; ch++) {
    __syncthreads();
    for (int e = tid; e < KC * PR * PC; e += 256) {
      const int c = e / (PR * PC), r = (e / PC) % PR, q = e % PC;
      patch[c][r][q] = xt[((size_t)(ch * KC + c) * H + r) * W + q];
    }
    __syncthreads();
    float d[6][4], h[6][4];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int q = 0; q < 4; q++) d[r][q] = patch[ci][4 * ty + r][2 * tx + q];
#pragma unroll
    for (int r = 0; r < 6; r++) {  // F(2,3) along the row: (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
      h[r][0] = d[r][0] - d[r][2], h[r][1] = d[r][1] + d[r][2], h[r][2] = d[r][2] - d[r][1], h[r][3] = d[r][1] - d[r][3];
    }
    float *vo = vb + (size_t)ch * VSZ + ci * NT + tile;
#pragma unroll
    for (int j = 0; j < 4; j++) {  // Bt6 down the column
      const float a0 = h[0][j], a1 = h[1][j], a2 = h[2][j], a3 = h[3][j], a4 = h[4][j], a5 = h[5][j];
      vo[(0 * 4 + j) * KC * NT] = 4.f * a0 - 5.f * a2 + a4;
      vo[(1 * 4 + j) * KC * NT] = -4.f * a1 - 4.f * a2 + a3 + a4;
      vo[(2 * 4 + j) * KC * NT] = 4.f * a1 - 4.f * a2 - a3 + a4;
      vo[(3 * 4 + j) * KC * NT] = -2.f * a1 - a2 + 2.f * a3 + a4;
      vo[(4 * 4 + j) * KC * NT] = 2.f * a1 - a2 - 2.f * a3 + a4;
      vo[(5 * 4 + j) * KC * NT] = 4.f * a1 - 5.f * a3 + a5;
    }
  }
}

// (2) the operand traffic of the GEMM kernel alone: 8 waves, 144 KB of LDS (one workgroup per CU), 48 chunks, stage
// k + 1 requested before stage k is waited for.  VFORM: 3 16-byte pieces per thread from V; else 6 dwords per thread
// from the input patch (the real kernel's addressing, clamped)
template <bool VFORM>
__global__ __launch_bounds__(512) void stream_operands(const float *__restrict__ src, float *__restrict__ sink, int ncb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // launch order: cout block fastest, the cout blocks of a tile block on ONE XCD (workgroup ids go round the 8 XCDs)
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int blk = (slot / ncb) * 8 + xcd, cob = slot % ncb;
  if (blk >= NBLK) return;
  const int tid = threadIdx.x, wave = tid >> 6;
  float acc = 0.f;
  if (VFORM) {
    const float *vb = src + (size_t)blk * NCHUNK * VSZ;
    auto issue = [&](int ch, int buf) {
#pragma unroll
      for (int p = 0; p < 3; p++)
        __builtin_amdgcn_global_load_lds((glb_ptr_t *)(vb + (size_t)ch * VSZ + (p * 512 + tid) * 4),
                                         (lds_ptr_t *)(lds + buf * VSZ + (p * 512 + wave * 64) * 4), 16, 0, 0);
    };
    issue(0, 0);
    for (int ch = 0; ch < NCHUNK; ch++) {
      if (ch + 1 < NCHUNK) issue(ch + 1, (ch + 1) & 1);
      if (ch + 1 < NCHUNK)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      acc += lds[(ch & 1) * VSZ + tid];
      __syncthreads();
    }
  } else {
    const int cb = blk % CB, rb = (blk / CB) % RB, t = blk / (CB * RB);
    const float *xt = src + (size_t)t * C * H * W + (size_t)(8 * rb) * W + 64 * cb;
    constexpr int PSZ = KC * PR * PC, PBUF = 6 * 512;
    auto issue = [&](int ch, int buf) {
#pragma unroll
      for (int p = 0; p < 6; p++) {
        int e = p * 512 + tid;
        e = e < PSZ ? e : 0;
        const int c = e / (PR * PC), r = (e / PC) % PR, q = e % PC;
        __builtin_amdgcn_global_load_lds((glb_ptr_t *)(xt + ((size_t)(ch * KC + c) * H + r) * W + q),
                                         (lds_ptr_t *)(lds + buf * PBUF + p * 512 + wave * 64), 4, 0, 0);
      }
    };
    issue(0, 0);
    for (int ch = 0; ch < NCHUNK; ch++) {
      if (ch + 1 < NCHUNK) issue(ch + 1, (ch + 1) & 1);
      if (ch + 1 < NCHUNK)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      acc += lds[(ch & 1) * PBUF + tid];
      __syncthreads();
    }
  }
  if (acc == 12345.678f) sink[id] = acc + cob;  // (never: keeps the reads alive)
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t nx = (size_t)TN * C * H * W, nv = (size_t)NBLK * NCHUNK * VSZ;
  float *x, *v, *sink;
  CHECK(hipMalloc(&x, nx * 4));
  CHECK(hipMalloc(&v, nv * 4));
  CHECK(hipMalloc(&sink, 1 << 20));
  std::vector<float> h(nx);
  unsigned s = 1u;
  for (float &f : h) s = s * 1664525u + 1013904223u, f = ((int)(s >> 9) % 2001 - 1000) / 1000.f;
  CHECK(hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("layer: %d tiles x %d channels x %d x %d in (%.1f MB), V %.1f MB (x %.2f)\n", TN, C, H, W, nx * 4 / 1e6, nv * 4 / 1e6,
         (double)nv / nx);
  auto timed = [&](const char *name, auto launch, double bytes) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      float ms = 0;
      (void)hipEventRecord(e0, 0);
      launch();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-64s %7.3f ms  (%.2f TB/s of %.2f GB)\n", name, best, bytes / best / 1e9, bytes / 1e9);
  };
  timed("pre-transform pass (input once -> V)", [&] { hipLaunchKernelGGL(pretransform, dim3(NBLK), dim3(256), 0, 0, x, v); },
        (double)(nx + nv) * 4);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_operands<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_operands<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  for (int ncb : {3, 12}) {
    const int grid = NBLK * ncb;
    char name[128];
    snprintf(name, sizeof name, "operand stream alone, %2d cout blocks: patch form (6 dwords)", ncb);
    timed(name, [&] { hipLaunchKernelGGL(stream_operands<false>, dim3(grid), dim3(512), 144 * 1024, 0, x, sink, ncb); },
          (double)grid * NCHUNK * KC * PR * PC * 4);
    snprintf(name, sizeof name, "operand stream alone, %2d cout blocks: V form (3 x 16 bytes)", ncb);
    timed(name, [&] { hipLaunchKernelGGL(stream_operands<true>, dim3(grid), dim3(512), 144 * 1024, 0, v, sink, ncb); },
          (double)grid * NCHUNK * VSZ * 4);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
