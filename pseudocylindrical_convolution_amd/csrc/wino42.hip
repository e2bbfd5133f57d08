// 3x3 stride-1 tile convolution by Winograd F(4x2, 3x3) on the fp32 matrix cores (round 4).
//
// wino.hip computes 2 x 2 outputs from a 4 x 4 input block with 16 multiply-adds per (input channel, output
// channel): 4 per output.  Here the vertical direction takes the larger minimal-filtering form F(4, 3) (six
// taps for four outputs), the horizontal one stays F(2, 3):
//
//     V = Bt6 d B4      (6 x 4 input block d, per input channel)       U = G6 g G4t   (per weight, packed once)
//     M[xi] = sum_ci U[xi][co][ci] * V[xi][ci][tile]   xi = 0..23: twenty-four independent [Cout x Cin] GEMMs
//     Y = At6 M A4      (4 x 2 outputs)
//
// 24 multiply-adds per 8 outputs = 3 per output: 1.33 x fewer matrix-core operations than F(2x2, 3x3) (3 x fewer
// than the direct form) with the same accumulator budget -- three quarters of the CU's register file -- the
// same LDS footprint and the same number of matrix instructions per workgroup, which now covers 64 couts x
// 64 tiles of 4 x 2 = 8 x 64 output pixels (wino.hip: 96 couts x 4 x 64 pixels).  Only the vertical transforms
// carry F(4, 3)'s larger constants (4, 5, 8, 1/24): the rounding error is about 3 x that of wino.hip (1.1e-5
// against float64 on unit-scale outputs), an order of magnitude below the codec's 1e-4 budget
// (tests/test_gpu_wino42.py).  F(4x4, 3x3) -- 2.25 per output -- would need 36 accumulator blocks, i.e. 64
// couts x 32 tiles per workgroup: 18 instead of 24 matrix instructions per chunk under the same transform / DMA
// / barrier cost per chunk (priced in DESIGN.md section 4: ~5 % faster at 3 x the error; not built).
//
// Structure = wino.hip's, re-tiled:
//   * a workgroup = 8 waves; wave w owns the GEMMs xi = 3w .. 3w+2: three 64 x 64 accumulator blocks of
//     2 x 2 MFMA tiles of 32 x 32 = 192 registers;
//   * input patch (4 channels x 10 x 66) -> LDS by LDS-DMA, double buffered;
//   * input transform LDS -> LDS into a double-buffered V[xi][ci][tile]: wave -> channel (wave & 3), lane ->
//     tile, waves 0-3 the rows i = 0..2 of Bt6, waves 4-7 the rows 3..5 (as wave-uniform coefficients, so
//     that all waves run the same instructions), in pieces placed between the wave's own MFMAs;
//   * transformed weights U: each wave streams its own slice through a PRIVATE three-stage LDS ring by
//     16-byte LDS-DMA and waits with s_waitcnt only;
//   * one workgroup barrier per 4 input channels (24 MFMAs per wave), main loop unrolled over six chunks
//     (patch / V buffers alternate, weight ring slots rotate: all compile-time);
//   * output transform: the 24 M[xi] of an output meet in LDS (eight half-rounds of a 16-cout x 32-tile block
//     of all xi, two 48 KB buffers, one barrier each); a thread turns one (cout, tile) pair per half-round
//     into 4 x 2 outputs and applies the epilogue (bias, PReLU, residual, trim, or the Dtow pixel shuffle).
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

#ifdef PCONV_W42_STAMP
// profiling build (tools/gpu_probe_wino42_stamps.py): s_memtime of one workgroup's phases, per wave:
// [start, prologue done, main loop done, way out done, barrier wait cycles of the steady chunks, steady chunks]
__device__ unsigned long long w42_stamps[8][6];
#define W42_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define W42_STAMP(var)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;
typedef const __attribute__((address_space(1))) char glb_bytes_t;

constexpr int kThreads = 512, kWaves = 8;
constexpr int XW = 3;                                // GEMMs (xi) per wave
constexpr int NXI = kWaves * XW;                     // 24 = 6 (vertical) x 4 (horizontal): xi = 4 i + j
constexpr int TX = 32, TY = 2;                       // Winograd tiles of a workgroup (a tile = 4 rows x 2 columns)
constexpr int OROWS = 4 * TY, OCOLS = 2 * TX;        // 8 x 64 output pixels
constexpr int CO = 64;                               // couts of a workgroup
constexpr int KC = 4;                                // input channels per stage (patch, V, weights)
constexpr int PR = OROWS + 2, PC = OCOLS + 2;        // 10 x 66 patch
constexpr int PSZ = KC * PR * PC;                    // 2640
constexpr int PLD = (PSZ + kThreads - 1) / kThreads; // 6 DMA dwords per thread
constexpr int PBUF = PLD * kThreads;                 // (the tail of the last piece re-reads element 0 into the slack)
constexpr int PRING = 2;                             // patch stages
constexpr int NT = TX * TY;                          // 64 tiles
constexpr int VSZ = NXI * KC * NT;                   // 6144: V[xi][ci][tile], double buffered
constexpr int USZ = KC * XW * CO;                    // 768: a wave's weight stage [ci][x][64]
constexpr int URING = 3;                             // weight stages per wave
constexpr int ULD = USZ / 4 / 64;                    // 3 16-byte DMA pieces per lane
constexpr int UNROLL = 6;                            // lcm(PRING, URING): chunks per unrolled group
constexpr int kStageFloats = PRING * PBUF + 2 * VSZ + kWaves * URING * USZ;  // 36864 floats = 144 KB
constexpr int kLdsFloats = kStageFloats + 256;       // + the cout block's bias [128] and PReLU slope [128]
constexpr int ESZ = NXI * 16 * 32;                   // one exchange half-round: all xi of 16 couts x 32 tiles
static_assert(2 * ESZ <= kStageFloats, "the two exchange buffers fit the stage memory");
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS of a CU");
static_assert(PSZ <= PBUF && USZ % 256 == 0, "stage sizes");
static_assert(PLD + 2 * ULD < 64, "vmcnt counts to 63");
static_assert(KC == 4 && NXI == 24, "transform: waves 0-3 take one channel of the stage each; three GEMMs per wave");

struct WView {
  long long ts, cs;
  int rs;
};

struct WEpilogue {
  const float *bias, *slope, *residual;
  const int32_t *col_limit;
  int npart, act, trim, d2w;
  WView vres;
};

// (cout, cin, 3, 3) -> U[cblock][wave][ci_pad][x][64] = (G6 g G4t)[xi = 3 wave + x], zero past cout / cin
__global__ void wino42_pack_kernel(const float *__restrict__ w, float *__restrict__ upk, int cout, int cin, int cin_pad,
                                   int cblocks) {
  const long long total = (long long)cblocks * cin_pad * CO;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % CO);
  const int ci = (int)((i / CO) % cin_pad);
  const int cb = (int)(i / ((long long)CO * cin_pad));
  const int co = cb * CO + j;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) g[a][b] = (co < cout && ci < cin) ? w[(((size_t)co * cin + ci) * 3 + a) * 3 + b] : 0.f;
  // rows: G6 (F(4,3)) over the vertical tap a
  float r[6][3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const float g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
    r[0][b] = g0 * 0.25f;
    r[1][b] = (g0 + g1 + g2) * (-1.f / 6.f);
    r[2][b] = (g0 - g1 + g2) * (-1.f / 6.f);
    r[3][b] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
    r[4][b] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
    r[5][b] = g2;
  }
#pragma unroll
  for (int a = 0; a < 6; a++) {
    // columns: G4 (F(2,3)) over the horizontal tap b
    const float u[4] = {r[a][0], (r[a][0] + r[a][1] + r[a][2]) * 0.5f, (r[a][0] - r[a][1] + r[a][2]) * 0.5f, r[a][2]};
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int xi = a * 4 + b;
      upk[((((size_t)cb * kWaves + xi / XW) * cin_pad + ci) * XW + (xi % XW)) * CO + j] = u[b];
    }
  }
}

// LDS operand reads of the matrix block, issued by hand (see wino.hip): `ds_read_b32 dst, base offset:imm`,
// counted waits -- before the MFMAs of a step only that step's four reads must have landed.
template <int OFF>
__device__ __forceinline__ float w42_lds_read(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int PENDING>
__device__ __forceinline__ void w42_wait(float (&a)[2], float (&b)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(PENDING));
}

// operands of step ST = (GEMM x, k-pair kp) of the chunk in weight slot U / V buffer VB
template <int U, int VB, int ST>
__device__ __forceinline__ void w42_read_step(float (&a)[2], float (&b)[2], unsigned abase, unsigned bbase) {
  constexpr int x = ST >> 1, kp = ST & 1;
  constexpr int aoff = (U * USZ + (kp * 2 * XW + x) * CO) * 4;
  constexpr int boff = (VB * VSZ + (x * KC + kp * 2) * NT) * 4;
  a[0] = w42_lds_read<aoff>(abase);
  a[1] = w42_lds_read<aoff + 128>(abase);
  b[0] = w42_lds_read<boff>(bbase);
  b[1] = w42_lds_read<boff + 128>(bbase);
}

// Half of the input transform V = Bt6 d B4 of a (channel, tile) pair -- three of the six rows of Bt6 -- in
// pieces the matrix block places between its MFMAs.  Which half is a property of the wave: it reads the five
// patch rows r0 .. r0 + 4 (r0 = 0 / 1) and combines their horizontal transforms t[a] with the wave-uniform
// coefficients cf[row][a] (rows 0-2 of Bt6 over t[0..4], rows 3-5 over t[1..5]).
struct W42Half {
  const float *tp;   // first of the five patch rows of the pair's block in patch stage 0 (row pitch PC)
  float *tv;         // slot of the half's first value in V buffer 0
  float cf[3][5];
};
struct W42Rows {
  float t[5][4];
};
template <int PB>
__device__ __forceinline__ void w42_half_load(const W42Half &h, W42Rows &w) {
  const float *p = h.tp + PB * PBUF;
#pragma unroll
  for (int a = 0; a < 5; a++) {
    const f32x2 lo = *reinterpret_cast<const f32x2 *>(p + a * PC), hi = *reinterpret_cast<const f32x2 *>(p + a * PC + 2);
    w.t[a][0] = lo.x, w.t[a][1] = lo.y, w.t[a][2] = hi.x, w.t[a][3] = hi.y;
  }
}
// horizontal F(2,3) transform of the five rows, in place: (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
__device__ __forceinline__ void w42_half_horizontal(W42Rows &w) {
#pragma unroll
  for (int a = 0; a < 5; a++) {
    const float d0 = w.t[a][0], d1 = w.t[a][1], d2 = w.t[a][2], d3 = w.t[a][3];
    w.t[a][0] = d0 - d2, w.t[a][1] = d1 + d2, w.t[a][2] = d2 - d1, w.t[a][3] = d1 - d3;
  }
}
// row R (0..2) of the half: sum_a cf[R][a] t[a][j], j = 0..3, stored to V[xi = 4 (3 half + R) + j]
template <int VBUF, int R>
__device__ __forceinline__ void w42_half_row(const W42Half &h, const W42Rows &w) {
  constexpr int XS = KC * NT;
  float *v = h.tv + VBUF * VSZ + R * 4 * XS;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float s = h.cf[R][4] * w.t[4][j];
    s = __builtin_fmaf(h.cf[R][3], w.t[3][j], s);
    s = __builtin_fmaf(h.cf[R][2], w.t[2][j], s);
    s = __builtin_fmaf(h.cf[R][1], w.t[1][j], s);
    s = __builtin_fmaf(h.cf[R][0], w.t[0][j], s);
    v[j * XS] = s;
  }
}

// The matrix block of a chunk: six steps (GEMM x, k-pair kp) of four MFMAs, operands of step s+1 read before
// the MFMAs of step s.  Between its MFMAs the wave transforms ITS half of a (channel, tile) pair of the next
// chunk (load after the first MFMA of step 0, the horizontal pass in step 1, one row of Bt6 in each of steps
// 2-4) and, in the steady state (IL), issues the chunk's DMA instructions: patch pieces in steps 0-1, weight
// pieces in step 5 (the stage this block reads is free once the operands of its last step have landed).
template <int U, int VB, int ST, bool IL, class PD, class WD>
__device__ __forceinline__ void w42_mma_steps(f32x16 (&acc)[XW][2][2], float (&a)[2][2], float (&b)[2][2], unsigned abase,
                                              unsigned bbase, const W42Half &h, W42Rows &rows, PD &pd, WD &wd) {
  constexpr int NST = 2 * XW;
  constexpr int X = ST >> 1, S = ST & 1;
  static_assert(NST == 6, "the transform pieces are placed by hand in six steps");
  if constexpr (ST + 1 < NST) w42_read_step<U, VB, ST + 1>(a[(ST + 1) & 1], b[(ST + 1) & 1], abase, bbase);
  w42_wait<(ST + 1 < NST) ? 4 : 0>(a[S], b[S]);
  __builtin_amdgcn_sched_barrier(0);
  using std::integral_constant;
  auto dma = [&](auto g_c) {  // DMA piece of gap g of this step, if it has one
    constexpr int G = decltype(g_c)::value;
    if constexpr (IL && ST <= 1 && ST * 3 + G < PLD && G < 3) {
      __builtin_amdgcn_sched_barrier(0);
      pd(integral_constant<int, ST * 3 + G>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (IL && ST == NST - 1 && G < ULD) {
      __builtin_amdgcn_sched_barrier(0);
      wd(integral_constant<int, G>{});
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  acc[X][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][0], b[S][0], acc[X][0][0], 0, 0, 0);
  if constexpr (ST == 0) {
    __builtin_amdgcn_sched_barrier(0);
    w42_half_load<VB ^ 1>(h, rows);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (ST == 1) {
    __builtin_amdgcn_sched_barrier(0);
    w42_half_horizontal(rows);
    __builtin_amdgcn_sched_barrier(0);
  }
  dma(integral_constant<int, 0>{});
  acc[X][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][0], b[S][1], acc[X][0][1], 0, 0, 0);
  dma(integral_constant<int, 1>{});
  acc[X][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][1], b[S][0], acc[X][1][0], 0, 0, 0);
  if constexpr (ST >= 2 && ST <= 4) {
    __builtin_amdgcn_sched_barrier(0);
    w42_half_row<VB ^ 1, ST - 2>(h, rows);
    __builtin_amdgcn_sched_barrier(0);
  }
  dma(integral_constant<int, 2>{});
  acc[X][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][1], b[S][1], acc[X][1][1], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (ST + 1 < NST) w42_mma_steps<U, VB, ST + 1, IL>(acc, a, b, abase, bbase, h, rows, pd, wd);
}

template <bool RES, bool D2W>
__global__ __launch_bounds__(kThreads) void wino42_conv3x3_kernel(
    const float *__restrict__ in, const float *__restrict__ upk, float *out, int cin, int cin_pad, int h,
    int w, int cout, int ho, int wo, int tiles_r, int tiles_c, int cblocks, WView vin, WView vout, WEpilogue ep) {
  W42_STAMP(st_start);
  extern __shared__ float lds[];
  float *Ps = lds, *Vs = lds + PRING * PBUF, *Us = lds + PRING * PBUF + 2 * VSZ;

  int b = blockIdx.x;
  const int cb = b % cblocks;
  b /= cblocks;
  const int trx = b % tiles_r;
  b /= tiles_r;
  const int tcx = b % tiles_c;
  const int t = b / tiles_c;
  const int r0 = trx * OROWS, c0 = tcx * OCOLS, cout0 = cb * CO;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  float *outp = out + (size_t)t * vout.ts;
  const int limit = ep.col_limit ? ep.col_limit[t % ep.npart] : wo;

  if (c0 >= limit) {
    // the tile lies entirely in the dead columns of this latitude band: zeros
    for (int e = tid; e < CO * OROWS * OCOLS; e += kThreads) {
      const int col = e % OCOLS, row = (e / OCOLS) % OROWS, co = cout0 + e / (OCOLS * OROWS);
      if (co < cout && r0 + row < ho && c0 + col < wo) {
        if (D2W)
          outp[(size_t)(co >> 2) * vout.cs + (size_t)(2 * (r0 + row) + ((co >> 1) & 1)) * vout.rs + 2 * (c0 + col) + (co & 1)] = 0.f;
        else
          outp[(size_t)co * vout.cs + (size_t)(r0 + row) * vout.rs + c0 + col] = 0.f;
      }
    }
    return;
  }

  const float *inp = in + (size_t)t * vin.ts;
  const int nchunk = cin_pad / KC;

  // ---- LDS-DMA: patch (all threads), weights (each wave its own slice) ----
  unsigned xoffs[PLD];
#pragma unroll
  for (int j = 0; j < PLD; j++) {
    int e = tid + j * kThreads;
    e = e < PSZ ? e : 0;
    const int pc = e % PC, pr = (e / PC) % PR, ci = e / (PC * PR);
    int ir = r0 + pr, ic = c0 + pc;
    ir = ir < h ? ir : h - 1;
    ic = ic < w ? ic : w - 1;
    xoffs[j] = (unsigned)(((long long)ci * vin.cs + (long long)ir * vin.rs + ic) * 4);  // bytes
  }
  const size_t xstep = (size_t)KC * vin.cs;
  auto patch_piece = [&](int chunk, int buf, int j) {
    unsigned xo = xoffs[j];
    asm volatile("" : "+v"(xo));
    glb_bytes_t *xb = (glb_bytes_t *)(inp + chunk * xstep);
    __builtin_amdgcn_global_load_lds((glb_ptr_t *)(xb + xo), (lds_ptr_t *)(Ps + buf * PBUF + j * kThreads + wave * 64), 4, 0,
                                     0);
  };
  auto issue_patch = [&](int chunk, int buf, bool guard) {
    if (guard && chunk >= nchunk) return;
#pragma unroll
    for (int j = 0; j < PLD; j++) patch_piece(chunk, buf, j);
  };
  // a weight stage = KC channels x the wave's three GEMMs x 64 floats = 3 KB contiguous in the packed
  // weights: three 16-byte LDS-DMA instructions
  const float *uw = upk + ((size_t)cb * kWaves + wave) * cin_pad * XW * CO;  // wave-uniform
  float *us_w = Us + wave * URING * USZ;
  auto weight_piece = [&](int chunk, auto j_c) {
    constexpr int j = decltype(j_c)::value;
    glb_bytes_t *src = (glb_bytes_t *)(uw + (size_t)chunk * USZ);  // (uniform)
    float *dst = us_w + (chunk % URING) * USZ;
    unsigned lo = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lo));
    __builtin_amdgcn_global_load_lds((glb_ptr_t *)(src + lo), (lds_ptr_t *)dst, 16, j * 1024, 0);
  };
  auto issue_weights = [&](int chunk, bool guard) {
    if (guard && chunk >= nchunk) return;
    static_assert(ULD == 3, "weight pieces are issued by name");
    weight_piece(chunk, std::integral_constant<int, 0>{});
    weight_piece(chunk, std::integral_constant<int, 1>{});
    weight_piece(chunk, std::integral_constant<int, 2>{});
  };

  // ---- input transform: wave -> channel (wave & 3) of the stage, lane -> tile; waves 0-3 rows 0-2 of Bt6,
  // waves 4-7 rows 3-5.  Bt6 = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
  const int wave4 = wave >> 2, tch = wave & 3;  // (uniform)
  const int tty = lane >> 5, ttx = lane & 31;
  const float *tp0 = Ps + (tch * PR + 4 * tty + wave4) * PC + 2 * ttx;  // rows wave4 .. wave4 + 4 of the pair's 6 x 4 block
  float *tv0 = Vs + (wave4 * 12 * KC + tch) * NT + lane;               // V[xi = 12 wave4][tch][tile]
  const W42Half half_t = {tp0, tv0,
                          {{wave4 ? -2.f : 4.f, wave4 ? -1.f : 0.f, wave4 ? 2.f : -5.f, wave4 ? 1.f : 0.f, wave4 ? 0.f : 1.f},
                           {wave4 ? 2.f : 0.f, wave4 ? -1.f : -4.f, wave4 ? -2.f : -4.f, 1.f, wave4 ? 0.f : 1.f},
                           {wave4 ? 4.f : 0.f, wave4 ? 0.f : 4.f, wave4 ? -5.f : -4.f, wave4 ? 0.f : -1.f, 1.f}}};
  auto transform_full = [&](int pbuf, int vbuf) {  // this wave's half, in one piece (stage 0)
    W42Rows rows;
    const float *p = tp0 + pbuf * PBUF;
#pragma unroll
    for (int a = 0; a < 5; a++) {
      const f32x2 lo = *reinterpret_cast<const f32x2 *>(p + a * PC), hi = *reinterpret_cast<const f32x2 *>(p + a * PC + 2);
      rows.t[a][0] = lo.x, rows.t[a][1] = lo.y, rows.t[a][2] = hi.x, rows.t[a][3] = hi.y;
    }
    w42_half_horizontal(rows);
    constexpr int XS = KC * NT;
    float *v = tv0 + vbuf * VSZ;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float s = half_t.cf[r][4] * rows.t[4][j];
        s = __builtin_fmaf(half_t.cf[r][3], rows.t[3][j], s);
        s = __builtin_fmaf(half_t.cf[r][2], rows.t[2][j], s);
        s = __builtin_fmaf(half_t.cf[r][1], rows.t[1][j], s);
        s = __builtin_fmaf(half_t.cf[r][0], rows.t[0][j], s);
        v[(r * 4 + j) * XS] = s;
      }
  };

  f32x16 acc[XW][2][2];
#pragma unroll
  for (int x = 0; x < XW; x++)
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[x][m][n][r] = 0.f;

  // ---- prologue ----
  float *Bt = lds + kStageFloats;  // bias [0..63], slope [128..191] of the cout block
  if (wave < 4) {
    const float *src = wave < 2 ? ep.bias : (ep.act == 1 ? ep.slope : nullptr);
    int co = cout0 + (wave & 1) * 64 + lane;
    co = co < cout ? co : cout - 1;
    if (src)
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)(src + co), (lds_ptr_t *)(Bt + wave * 64), 4, 0, 0);
    else
      Bt[wave * 64 + lane] = 0.f;
  }
  issue_patch(0, 0, true);
#pragma unroll
  for (int k = 0; k < URING; k++) issue_weights(k, true);
  issue_patch(1, 1, true);
  __builtin_amdgcn_s_waitcnt(0);  // (vmcnt(0) among others)
  __syncthreads();
  transform_full(0, 0);

  // per-lane LDS byte addresses of the operand fragments (see w42_read_step): A = weights
  // [stage][ci = 2 kp + half][x][64], B = V[vbuf][xi = 3 wave + x][ci = 2 kp + half][tile]
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);  // low half of the flat address = LDS offset
  const unsigned abase = lds0 + (unsigned)((us_w - lds) + half * XW * CO + l31) * 4u;
  const unsigned bbase = lds0 + (unsigned)((Vs - lds) + ((wave * XW) * KC + half) * NT + l31) * 4u;

  // One chunk (KC input channels).  US = chunk % 3 (weight ring slot) and VB = chunk & 1 (patch / V buffer) are
  // compile-time (the loop below is unrolled over six chunks).  DMA issue order of a wave, one patch stage and one
  // weight stage per chunk:  ... patch(chunk+1), weights(chunk+2) | patch(chunk+2), weights(chunk+3).  On arrival
  // patch(chunk+1) -- and with it everything older, weights(chunk) included -- must have landed; weights(chunk+2),
  // issued a moment ago, stays in flight (counted wait).  Then everybody's: V(chunk) is complete (lgkmcnt(0): this
  // thread's LDS writes) and the MFMAs of chunk-1, last readers of V's other buffer, are done.
#ifdef PCONV_W42_STAMP
  unsigned long long st_bar = 0, st_n = 0;
#endif
  W42_STAMP(st_prologue);
  auto body = [&](auto us_c, auto vb_c, auto steady_c, int chunk) {
    constexpr int US = decltype(us_c)::value;
    constexpr int vb = decltype(vb_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    W42_STAMP(tb0);
    if (STEADY)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(ULD) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef PCONV_W42_STAMP
    if (STEADY) st_bar += __builtin_readcyclecounter() - tb0, st_n += 1;
#endif
    if (!STEADY) issue_patch(chunk + 2, vb, true);  // (that stage was read by the transform of this chunk, before the barrier)
    float a[2][2], bv[2][2];
    W42Rows rows;
    w42_read_step<US, vb, 0>(a[0], bv[0], abase, bbase);
    // (the last piece holds elements 2560 .. 2639 of the stage: waves 0-1; the others would only re-read element 0)
    auto pd = [&](auto j_c) {
      constexpr int J = decltype(j_c)::value;
      if (J * kThreads + wave * 64 >= PSZ) return;  // (uniform)
      patch_piece(chunk + 2, vb, J);
    };
    auto wd = [&](auto j_c) { weight_piece(chunk + URING, j_c); };
    w42_mma_steps<US, vb, 0, STEADY>(acc, a, bv, abase, bbase, half_t, rows, pd, wd);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (this wave's reads of the weight stage are complete -- their values fed the MFMAs: refill it)
    if (!STEADY) issue_weights(chunk + URING, true);
  };
  using std::integral_constant;
  auto group = [&](auto steady_c, int c6) {
    body(integral_constant<int, 0>{}, integral_constant<int, 0>{}, steady_c, c6);
    body(integral_constant<int, 1>{}, integral_constant<int, 1>{}, steady_c, c6 + 1);
    body(integral_constant<int, 2>{}, integral_constant<int, 0>{}, steady_c, c6 + 2);
    body(integral_constant<int, 0>{}, integral_constant<int, 1>{}, steady_c, c6 + 3);
    body(integral_constant<int, 1>{}, integral_constant<int, 0>{}, steady_c, c6 + 4);
    body(integral_constant<int, 2>{}, integral_constant<int, 1>{}, steady_c, c6 + 5);
  };
  // nchunk is a multiple of 6 (cin % 24 == 0).  Two plain loops -- steady state, tail (see wino.hip).
  // (the prologue leaves the steady state's invariants behind -- patch(0), patch(1) and the weight stages 0..2
  // landed, V(0) written -- so the first group is a steady one too; only the last group, where the streams end,
  // takes the guarded form)
  int c6 = 0;
#pragma unroll 1
  for (; c6 + 2 * UNROLL <= nchunk; c6 += UNROLL) group(integral_constant<bool, true>{}, c6);
#pragma unroll 1
  for (; c6 < nchunk; c6 += UNROLL) group(integral_constant<bool, false>{}, c6);
  W42_STAMP(st_loop);
  __syncthreads();  // all MFMAs done: the stage memory becomes the exchange buffer

  // ---- output transform + epilogue ----
  // Eight half-rounds (m = cout tile, n = tile row, hq = half of the cout tile): the 24 M[xi] of 16 couts x 32
  // tiles through one of two exchange buffers (a half-round's readers are past their reads when they arrive at
  // the next one's barrier: one barrier per half-round).  A thread finishes ONE (cout, tile) pair per half-round
  // (d2w: the two couts of a pair take turns).  The residual is requested two half-rounds ahead.
  const int act = ep.act;
  const int trim_at = ep.trim ? limit : wo;
  const float *resp = RES ? ep.residual + (size_t)t * ep.vres.ts : nullptr;
  const int co16 = tid >> 5, etx = tid & 31;  // the thread's pair inside a half-round: cout row, tile column
  const int ocol = c0 + 2 * etx;
  struct ResBlock {
    f32x2 v[4];
  };
  auto load_res = [&](int k) {
    ResBlock rb = {{{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}};
    if (RES && k < 8) {
      const int m = k >> 2, n = (k >> 1) & 1, hq = k & 1;
      const int oc = ocol < wo ? ocol : wo - 2;
      int co = cout0 + m * 32 + hq * 16 + co16;
      co = co < cout ? co : cout - 1;
#pragma unroll
      for (int a2 = 0; a2 < 4; a2++) {
        int rr = r0 + 4 * n + a2;
        rr = rr < ho ? rr : ho - 1;
        rb.v[a2] = *reinterpret_cast<const f32x2 *>(resp + (size_t)co * ep.vres.cs + (size_t)rr * ep.vres.rs + oc);
      }
    }
    return rb;
  };
  ResBlock rq[2] = {load_res(0), load_res(1)};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int m = k >> 2, n = (k >> 1) & 1, hq = k & 1;
    float *Es = lds + (k & 1) * ESZ;  // [xi][16 couts][32 tiles]
    const ResBlock rcur = rq[k & 1];
    rq[k & 1] = load_res(k + 2);
    // accumulator register r of lane (half, l31) = cout row (r & 3) + 8 (r >> 2) + 4 half of the 32-cout tile:
    // registers 8 hq .. 8 hq + 7 are the rows 16 hq .. 16 hq + 15
#pragma unroll
    for (int x = 0; x < XW; x++) {
      float *e = Es + ((wave * XW + x) * 16 + 4 * half) * 32 + l31;
#pragma unroll
      for (int rl = 0; rl < 8; rl++) e[((rl & 3) + 8 * (rl >> 2)) * 32] = acc[x][m][n][8 * hq + rl];
    }
    // (this thread's exchange stores are out; the barrier must not wait for the global stores of the half-round
    // before -- __syncthreads() waits for vmcnt(0), a store -> barrier chain of eight memory round trips)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float mm[NXI];
#pragma unroll
    for (int xi = 0; xi < NXI; xi++) mm[xi] = Es[(xi * 16 + co16) * 32 + etx];
    const int row = m * 32 + hq * 16 + co16;  // cout inside the block
    const int co = cout0 + row;
    const int orow = r0 + 4 * n;
    if (co < cout && ocol < wo) {
      // Y = At6 M A4, M[i][j] = mm[4 i + j]: horizontal (1 1 1 0; 0 1 -1 -1), then vertical
      // At6 = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
      float s[6][2];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        s[i][0] = mm[4 * i] + mm[4 * i + 1] + mm[4 * i + 2];
        s[i][1] = mm[4 * i + 1] - mm[4 * i + 2] - mm[4 * i + 3];
      }
      float y[4][2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const float p12 = s[1][c] + s[2][c], m12 = s[1][c] - s[2][c];
        const float p34 = s[3][c] + s[4][c], m34 = s[3][c] - s[4][c];
        y[0][c] = s[0][c] + p12 + p34;
        y[1][c] = __builtin_fmaf(2.f, m34, m12);
        y[2][c] = __builtin_fmaf(4.f, p34, p12);
        y[3][c] = __builtin_fmaf(8.f, m34, m12) + s[5][c];
      }
      const float bco = Bt[row], sl = Bt[128 + row];
#pragma unroll
      for (int a2 = 0; a2 < 4; a2++) {
        if (orow + a2 >= ho) continue;
        float y0 = y[a2][0] + bco, y1 = y[a2][1] + bco;
        if (act == 1) {
          y0 = y0 < 0 ? y0 * sl : y0;
          y1 = y1 < 0 ? y1 * sl : y1;
        }
        if (RES) {
          y0 = rcur.v[a2].x + y0;
          y1 = rcur.v[a2].y + y1;
        }
        if (!D2W) {
          if (ocol >= trim_at) y0 = 0.f;
          if (ocol + 1 >= trim_at) y1 = 0.f;
          float *q = outp + (size_t)co * vout.cs + (size_t)(orow + a2) * vout.rs + ocol;
          f32x2 yv = {y0, y1};
          *reinterpret_cast<f32x2 *>(q) = yv;  // (wo is even: a 4 x 2 block never straddles the right edge)
        } else {
          // Dtow by the store: cout co -> channel co >> 2, row 2 r + ((co >> 1) & 1), column 2 c + (co & 1)
          float *q = outp + (size_t)(co >> 2) * vout.cs + (size_t)(2 * (orow + a2) + ((co >> 1) & 1)) * vout.rs + 2 * ocol +
                     (co & 1);
          q[0] = y0;
          q[2] = y1;
        }
      }
    }
  }
#ifdef PCONV_W42_STAMP
  if (blockIdx.x == gridDim.x / 2 + 1 && lane == 0) {
    unsigned long long *o = w42_stamps[wave];
    o[0] = st_start, o[1] = st_prologue, o[2] = st_loop, o[3] = __builtin_readcyclecounter(), o[4] = st_bar, o[5] = st_n;
  }
#endif
}

inline WView dense_view(int c, int h, int w) { return {(long long)c * h * w, (long long)h * w, w}; }
inline WView view_at(const long long *views, int i, int c, int h, int w) {
  if (!views) return dense_view(c, h, w);
  return {views[3 * i], views[3 * i + 1], (int)views[3 * i + 2]};
}
inline bool view_ok(const WView &v, int c, int h, int w) {
  return v.rs >= w && v.cs >= (long long)(h - 1) * v.rs + w && v.ts >= (long long)(c - 1) * v.cs + (long long)(h - 1) * v.rs + w;
}

}  // namespace

#ifdef PCONV_W42_STAMP
extern "C" int pconv_wino42_read_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(w42_stamps), sizeof(w42_stamps)) == hipSuccess ? 0 : 1;
}
#endif

// floats of the packed F(4x2, 3x3) weights of a (cout, cin, 3, 3) layer
extern "C" long long pconv_wino42_packed_size(int cout, int cin) {
  const int cblocks = (cout + CO - 1) / CO, cin_pad = (cin + KC - 1) / KC * KC;
  return (long long)cblocks * NXI * cin_pad * CO;
}

extern "C" int pconv_wino42_pack_weight(const float *w, float *packed, int cout, int cin, void *stream) {
  PCONV_REQUIRE(w && packed && cout > 0 && cin > 0, "wino42_pack: bad argument");
  const int cblocks = (cout + CO - 1) / CO, cin_pad = (cin + KC - 1) / KC * KC;
  const long long total = (long long)cblocks * cin_pad * CO;
  hipLaunchKernelGGL(wino42_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w, packed,
                     cout, cin, cin_pad, cblocks);
  PCONV_LAUNCH_CHECK("wino42_pack_weight");
  return PCONV_OK;
}

// 1 when pconv_conv3x3_wino42 takes the layer: 3x3 stride 1, at least 4 output rows, even output width (a last
// tile row that is not whole is computed from clamped patch rows and not stored), cin a multiple of 24 (six-chunk
// unrolled main loop), cout >= 32 (d2w: a multiple of 4)
extern "C" int pconv_wino42_supported(int cin, int h, int w, int cout, int d2w) {
  if (h < 6 || w < 4 || ((w - 2) & 1)) return 0;
  if (cin < 24 || cin % 24 || cout < 32) return 0;
  if (d2w && (cout & 3)) return 0;
  return 1;
}

extern "C" int pconv_conv3x3_wino42(const float *in, const float *packed_u, const float *bias, float *out, int tn, int cin,
                                    int h, int w, int cout, int act, const float *slope, const int32_t *col_limit,
                                    int npart, const float *residual, int trim, int d2w, const long long *views,
                                    void *stream) {
  PCONV_REQUIRE(in && packed_u && out, "conv3x3_wino42: null pointer");
  PCONV_REQUIRE(pconv_wino42_supported(cin, h, w, cout, d2w), "conv3x3_wino42: unsupported shape %d x %d x %d -> %d", cin, h,
                w, cout);
  PCONV_REQUIRE(act == 0 || (act == 1 && slope), "conv3x3_wino42: bad activation %d", act);
  PCONV_REQUIRE(!d2w || (!residual && !trim), "conv3x3_wino42: depth-to-width takes no residual / trim");
  PCONV_REQUIRE(!col_limit || npart > 0, "conv3x3_wino42: col_limit needs npart");
  PCONV_REQUIRE(!trim || col_limit, "conv3x3_wino42: trim needs col_limit");
  PCONV_REQUIRE(residual != out, "conv3x3_wino42: residual must not alias the output");
  const int ho = h - 2, wo = w - 2;
  const int oc = d2w ? cout / 4 : cout, oh = d2w ? 2 * ho : ho, ow = d2w ? 2 * wo : wo;
  const WView vin = view_at(views, 0, cin, h, w), vout = view_at(views, 1, oc, oh, ow);
  const WEpilogue ep = {bias, slope, residual, col_limit, npart, act, trim, d2w, view_at(views, 2, cout, ho, wo)};
  PCONV_REQUIRE(view_ok(vin, cin, h, w) && view_ok(vout, oc, oh, ow) && (!residual || view_ok(ep.vres, cout, ho, wo)),
                "conv3x3_wino42: strides overlap");
  PCONV_REQUIRE(((long long)(KC - 1) * vin.cs + (long long)(h - 1) * vin.rs + w) * 4 < (1LL << 32),
                "conv3x3_wino42: input channel stride too large for 32-bit byte offsets inside a chunk");
  // float2 accesses: rows of the output / residual views start on even element offsets
  PCONV_REQUIRE(d2w || (vout.rs % 2 == 0 && vout.cs % 2 == 0 && vout.ts % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0),
                "conv3x3_wino42: output rows must be 8-byte aligned");
  PCONV_REQUIRE(!residual || (ep.vres.rs % 2 == 0 && ep.vres.cs % 2 == 0 && ep.vres.ts % 2 == 0 &&
                              (reinterpret_cast<uintptr_t>(residual) & 7) == 0),
                "conv3x3_wino42: residual rows must be 8-byte aligned");
  const int tiles_r = (ho + OROWS - 1) / OROWS, tiles_c = (wo + OCOLS - 1) / OCOLS;
  const int cblocks = (cout + CO - 1) / CO, cin_pad = (cin + KC - 1) / KC * KC;
  const long long grid = (long long)tn * tiles_r * tiles_c * cblocks;
  PCONV_REQUIRE(grid > 0 && grid <= 0x7fffffffLL, "conv3x3_wino42: grid %lld out of range", grid);
  const size_t smem = (size_t)kLdsFloats * sizeof(float);
  // one instantiation per kind of layer: 0 plain, 1 residual, 2 depth-to-width
  using kernel_t = decltype(&wino42_conv3x3_kernel<false, false>);
  static const kernel_t kernels[3] = {wino42_conv3x3_kernel<false, false>, wino42_conv3x3_kernel<true, false>,
                                      wino42_conv3x3_kernel<false, true>};
  const int kind = d2w ? 2 : (residual ? 1 : 0);
  {
    static std::atomic<unsigned long long> raised[3];
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    const unsigned long long bit = 1ULL << (device & 63);
    if (!(raised[kind].load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernels[kind]),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) {
        pconv_set_error("conv3x3_wino42: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e));
        return PCONV_ELAUNCH;
      }
      raised[kind].fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kernels[kind], dim3((unsigned)grid), dim3(kThreads), smem, as_stream(stream), in, packed_u, out, cin,
                     cin_pad, h, w, cout, ho, wo, tiles_r, tiles_c, cblocks, vin, vout, ep);
  PCONV_LAUNCH_CHECK("conv3x3_wino42");
  return PCONV_OK;
}
