// 3x3 stride-1 tile convolution by Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// The 3x3 stride-1 layers are 80 % of the codec's convolution time (model_zoo_v2.py:41-45,83-86,
// 158-164: ResidualBlock conv2, ResidualBlockV2 conv1/2, ResidualBlockDown/Up conv2, ResidualBlockUp
// conv1).  The direct implicit GEMM (conv.hip) spends 9 multiply-adds per input channel, output channel
// and pixel; the minimal-filtering form spends 16 per 2x2 output block = 4 per pixel:
//
//     V = Bt d B   (4x4 input block d, per input channel)        U = G g Gt   (per weight, packed once)
//     M[xi] = sum_ci U[xi][co][ci] * V[xi][ci][tile]   xi = 0..15: sixteen independent [Cout x Cin] GEMMs
//     Y = At M A   (2x2 outputs)
//
// i.e. 2.25 x fewer matrix-core operations for the same convolution -- what cuDNN picks for the
// reference's fp32 3x3 layers too.  Results differ from the k-ascending fmaf chain of conv.hip by
// rounding only (~1e-6 relative; the north-star bound is 1e-4); conv.hip stays the bit-exact form
// (PCONV_CONV3X3=direct) and takes everything this kernel does not: stride 2, gates, odd sizes, and any
// cin that is not a multiple of 16 or cout < 32 (pconv_wino_supported is the single source of truth).
//
// LDS contract: the kernel's dynamic LDS must start at LDS byte 0 (no static __shared__ in this
// translation unit's kernel): wino_store_round splits the second exchange buffer's address between
// M0[15:0] and the 16-bit immediate of ds_write_addtid assuming lds0 == 0; the kernel traps otherwise.
//
// Mapping.  A workgroup = 8 waves = 96 couts x 64 Winograd tiles (2 tile rows x 32 tile columns = 4 x 64
// output pixels).  Wave w owns the GEMMs xi = 2w, 2w+1: two 96 x 64 accumulator blocks of 3 x 2 MFMA tiles
// of 32 x 32 = 192 registers, so the sixteen GEMMs use three quarters of the CU's register file as
// accumulators -- the pixel tile cannot be larger, which is why a workgroup takes 96 and not 192 couts --
// and a wave still has ~60 registers for operands in flight (two waves per SIMD, 256 registers each;
// the first version, sixteen waves of one GEMM and 128 registers, had none: 207 vs 214-224 TFLOP/s).
//   * input patch (4 channels x 6 x 66) -> LDS by LDS-DMA, double buffered;
//   * input transform LDS -> LDS into a double-buffered V[xi][ci][tile]: every wave does half of one
//     (channel, tile) pair per thread (waves 0-3 rows 0-1 of Bt d, waves 4-7 rows 2-3), in pieces placed
//     BETWEEN its own MFMAs (wino_mma_steps): next to a wave that streams fp32 MFMAs its SIMD partner gets
//     about one vector / LDS instruction per matrix instruction issued, so a transform in front of the
//     matrix block made the two waves of a SIMD take turns (in-kernel stamps: 4 100 cycles per chunk for
//     3 072 cycles of MFMAs; 3 670 now);
//   * transformed weights U[xi]: nobody but the wave that owns xi reads them, so each wave streams its own
//     slice through a PRIVATE four-stage LDS ring by 16-byte LDS-DMA and waits with s_waitcnt only; the DMA
//     instructions of a chunk also sit between MFMAs, addressed as scalar base + 32-bit lane offset;
//   * one workgroup barrier per 4 input channels (24 MFMAs per wave), main loop unrolled over four chunks
//     (ring slots are compile-time constants);
//   * output transform: the sixteen M[xi] of an output meet in LDS (one 32-cout x 32-tile block of all
//     xi per round, 64 KB, two buffers), each thread turns (cout, tile) pairs into 2 x 2 outputs and applies
//     the epilogue of the layer (bias, PReLU, residual, trim, or the Dtow pixel shuffle), stored as float2 /
//     float4 runs; no load of the way out sits behind a branch or a store (bias / slope in an LDS table,
//     residual requested two rounds ahead with counted waits: the kind of layer is a template parameter).
//     (Measured alternative: three rounds with 16-byte writes / 8-byte reads -- half the LDS instructions
//     -- is not faster: 5.04 vs 5.02 ms.)
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

#ifdef PCONV_WINO_STAMP
// profiling build (tools/gpu_probe_wino_stamps.py): cycles (s_memtime) the waves of one workgroup spend in
// the phases of a steady-state chunk: [wave][barrier wait, head (patch DMA issue), matrix block, weight DMA
// issue, -, chunks]
__device__ unsigned long long wino_stamps[8][6];
#define WINO_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define WINO_STAMP(var)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;
typedef const __attribute__((address_space(1))) char glb_bytes_t;

constexpr int kThreads = 512, kWaves = 8;
constexpr int XW = 2;                                // GEMMs (xi) per wave
#ifdef PCONV_WINO_FLAT
// (r6) second build of this file (wino_flat.hip): one tile row of 64 tiles = 2 x 128 output pixels per workgroup, for
// the 2-row REMAINDERS of the row split (PCONV.tile_conv2d): the 4-row workgroup ran them half empty
constexpr int TX = 64, TY = 1;
#else
constexpr int TX = 32, TY = 2;                       // Winograd tiles of a workgroup
#endif
constexpr bool kFlat = TY == 1;
constexpr int OROWS = 2 * TY, OCOLS = 2 * TX;        // 4 x 64 output pixels (flat: 2 x 128)
constexpr int CO = 96;                               // couts of a workgroup
constexpr int KC = 4;                                // input channels per stage (patch, V, weights)
constexpr int PR = OROWS + 2, PC = OCOLS + 2;        // 6 x 66 patch
constexpr int PSZ = KC * PR * PC;                    // 1584
constexpr int PLD = (PSZ + kThreads - 1) / kThreads; // 4 DMA dwords per thread
constexpr int PBUF = PLD * kThreads;                 // every wave issues all PLD pieces (the last one re-reads
                                                     // element 0 past PSZ): uniform DMA counts for s_waitcnt vmcnt(n)
constexpr int PRING = 2;                             // patch stages
constexpr int VSZ = 16 * KC * TX * TY;               // 4096: V[xi][ci][tile], double buffered
constexpr int USZ = KC * XW * CO;                    // 768: a wave's weight stage [ci][xi & 1][96]
constexpr int URING = 4;                             // weight stages per wave
constexpr int ULD = USZ / 4 / 64;                    // 3 16-byte DMA pieces per lane
constexpr int kStageFloats = PRING * PBUF + 2 * VSZ + kWaves * URING * USZ;  // 36864 floats = 144 KB
constexpr int kLdsFloats = kStageFloats + 256;  // + the cout block's bias [128] and PReLU slope [128]
constexpr int ESZ = 16 * 32 * 32;                    // one exchange round of the output transform
static_assert(2 * ESZ <= kStageFloats, "the two exchange buffers fit the stage memory");
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS of a CU");
static_assert(PSZ <= PBUF && PBUF % 2 == 0, "patch buffer");
static_assert(USZ % 256 == 0, "weight stage = whole 16-byte DMA instructions");
static_assert(PLD + 2 * ULD < 64, "vmcnt counts to 63");
static_assert(PLD <= 5 && ULD <= 4, "DMA pieces of a chunk fit the gaps of the matrix block (a fifth patch piece: step 1)");
static_assert(URING == 4 && PRING == 2, "the main loop is unrolled over four chunks: ring slots are compile-time");
static_assert(XW == 2, "the way out stores a wave's two GEMMs by hand");
static_assert(KC == 4 && kWaves * XW == 16, "transform: waves 0-3 take one channel of the stage each; two GEMMs per wave");

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

// (cout, cin, 3, 3) -> U[cblock][xi / 2][ci_pad][xi & 1][96] = (G g Gt)[xi], zero past cout / cin
// (the two GEMMs of a wave interleaved per channel: a stage of KC channels is one contiguous 3 KB run)
__global__ void wino_pack_kernel(const float *__restrict__ w, float *__restrict__ upk, int cout, int cin, int cin_pad,
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
  float r[4][3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    r[0][b] = g[0][b];
    r[1][b] = (g[0][b] + g[1][b] + g[2][b]) * 0.5f;
    r[2][b] = (g[0][b] - g[1][b] + g[2][b]) * 0.5f;
    r[3][b] = g[2][b];
  }
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const float u[4] = {r[a][0], (r[a][0] + r[a][1] + r[a][2]) * 0.5f, (r[a][0] - r[a][1] + r[a][2]) * 0.5f, r[a][2]};
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int xi = a * 4 + b;
      upk[((((size_t)cb * kWaves + xi / XW) * cin_pad + ci) * XW + (xi % XW)) * CO + j] = u[b];
    }
  }
}

// LDS operand reads of the matrix block, issued by hand: `ds_read_b32 dst, base offset:imm` with the whole
// offset (ring slot, GEMM, k-pair, fragment: compile-time) in the 16-bit immediate, and COUNTED waits --
// before the MFMAs of a step only that step's five reads must have landed, the five of the next step,
// issued after them, stay in flight (LDS returns in order).  Left to itself the compiler waits for
// lgkmcnt(0) right after every read: with two waves per SIMD the read latency would sit on the matrix pipe.
template <int OFF>
__device__ __forceinline__ float wino_lds_read(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int PENDING>
__device__ __forceinline__ void wino_wait(float (&a)[3], float (&b)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]) : "n"(PENDING));
}

// operands of step ST = (GEMM x, k-pair kp) of the chunk in weight slot U / V buffer VB
template <int U, int VB, int ST>
__device__ __forceinline__ void wino_read_step(float (&a)[3], float (&b)[2], unsigned abase, unsigned bbase) {
  constexpr int x = ST >> 1, kp = ST & 1;
  constexpr int aoff = (U * USZ + (kp * 2 * XW + x) * CO) * 4;
  constexpr int boff = (VB * VSZ + (x * KC + kp * 2) * (TX * TY)) * 4;
  a[0] = wino_lds_read<aoff>(abase);
  a[1] = wino_lds_read<aoff + 128>(abase);
  a[2] = wino_lds_read<aoff + 256>(abase);
  b[0] = wino_lds_read<boff>(bbase);
  b[1] = wino_lds_read<boff + 128>(bbase);
}

// Half of the input transform V = Bt d B of a (channel, tile) pair -- two of the four rows of Bt d
// (d0 - d2, d1 + d2 | d2 - d1, d1 - d3) and their columns -- in three pieces that the matrix block below
// places between its MFMAs.  Which half is a property of the wave: row offsets rp / rq / rs (floats) and the
// sign sg are wave-uniform, so that every wave runs the same instructions:
//   first row = d[rp] - d[rq], second row = d[1] + sg * d[rs]   (an exact product: the same bits as d1 +- d)
struct WinoHalf {
  const float *tp;  // the pair's 4 x 4 block in patch stage 0 (row pitch PC)
  float *tv;        // slot of the half's first value in V buffer 0: V[xi][ci][tile], xi = 8 * half ..
  int rp, rq, rs;
  float sg;
};
struct WinoRows {
  f32x2 p[2], q[2], r[2], s[2];
};
template <int PB>
__device__ __forceinline__ void wino_half_load(const WinoHalf &h, WinoRows &w) {
  const float *t = h.tp + PB * PBUF;
  w.p[0] = *reinterpret_cast<const f32x2 *>(t + h.rp), w.p[1] = *reinterpret_cast<const f32x2 *>(t + h.rp + 2);
  w.q[0] = *reinterpret_cast<const f32x2 *>(t + h.rq), w.q[1] = *reinterpret_cast<const f32x2 *>(t + h.rq + 2);
  w.r[0] = *reinterpret_cast<const f32x2 *>(t + PC), w.r[1] = *reinterpret_cast<const f32x2 *>(t + PC + 2);
  w.s[0] = *reinterpret_cast<const f32x2 *>(t + h.rs), w.s[1] = *reinterpret_cast<const f32x2 *>(t + h.rs + 2);
}
// row J (0 / 1) of the half: its four columns (t0 - t2, t1 + t2, t2 - t1, t1 - t3)
template <int J>
__device__ __forceinline__ void wino_half_math(const WinoHalf &h, const WinoRows &w, float (&o)[4]) {
  float t[4];
  if (J == 0) {
    t[0] = w.p[0].x - w.q[0].x, t[1] = w.p[0].y - w.q[0].y, t[2] = w.p[1].x - w.q[1].x, t[3] = w.p[1].y - w.q[1].y;
  } else {
    t[0] = __builtin_fmaf(h.sg, w.s[0].x, w.r[0].x), t[1] = __builtin_fmaf(h.sg, w.s[0].y, w.r[0].y);
    t[2] = __builtin_fmaf(h.sg, w.s[1].x, w.r[1].x), t[3] = __builtin_fmaf(h.sg, w.s[1].y, w.r[1].y);
  }
  o[0] = t[0] - t[2], o[1] = t[1] + t[2], o[2] = t[2] - t[1], o[3] = t[1] - t[3];
}
template <int VBUF, int J>
__device__ __forceinline__ void wino_half_store(const WinoHalf &h, const float (&o)[4]) {
  constexpr int XS = KC * TX * TY;
  float *v = h.tv + VBUF * VSZ + J * 4 * XS;
#pragma unroll
  for (int j = 0; j < 4; j++) v[j * XS] = o[j];
}

// The matrix block of a chunk: four steps (GEMM x, k-pair kp) of six MFMAs, operands of step s+1 read before
// the MFMAs of step s.  Between its MFMAs the wave transforms ITS half of a (channel, tile) pair of the next
// chunk: a vector or LDS instruction costs nothing in the shadow of the wave's own matrix instruction, while
// in front of the block (this kernel's first form: waves 0-3 transformed, then multiplied) the same
// instructions were issued one per matrix instruction of the SIMD partner -- ~2 000 cycles for 37
// instructions (in-kernel stamps, profiles/round3_wino_stamps.txt) -- and the partner then waited at the
// barrier for the late wave's matrix block: the two waves of a SIMD took turns instead of sharing the pipe.
// IL: the chunk's DMA instructions sit there too, one per gap -- the patch pieces (`pd(j)`) in step 0, the
// weight pieces (`wd(j)`: the stage this block reads, free once the operands of its last step have landed)
// in step 3; issued in front of / behind the block they ran while neither wave of the SIMD had an MFMA ready.
template <int U, int VB, int ST, bool IL, class PD, class WD>
__device__ __forceinline__ void wino_mma_steps(f32x16 (&acc)[XW][3][2], float (&a)[2][3], float (&b)[2][2], unsigned abase,
                                               unsigned bbase, const WinoHalf &h, WinoRows &rows, PD &pd, WD &wd) {
  constexpr int NST = 2 * XW;
  constexpr int X = ST >> 1, S = ST & 1;
  static_assert(NST == 4, "the transform pieces are placed by hand in four steps");
  if constexpr (ST + 1 < NST) wino_read_step<U, VB, ST + 1>(a[(ST + 1) & 1], b[(ST + 1) & 1], abase, bbase);
  wino_wait<(ST + 1 < NST) ? 5 : 0>(a[S], b[S]);
  __builtin_amdgcn_sched_barrier(0);
  float o[4];
  acc[X][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][0], b[S][0], acc[X][0][0], 0, 0, 0);
  if constexpr (ST == 0) {
    __builtin_amdgcn_sched_barrier(0);
    wino_half_load<VB ^ 1>(h, rows);
    __builtin_amdgcn_sched_barrier(0);
  }
  auto dma = [&](auto j_c) {  // DMA piece j of this step, if it has one
    constexpr int J = decltype(j_c)::value;
    if constexpr (IL && ST == 0 && J < PLD) {
      __builtin_amdgcn_sched_barrier(0);
      pd(j_c);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (IL && ST == 1 && J == 0 && PLD > 4) {  // (flat build: 2080 patch elements, a fifth piece)
      __builtin_amdgcn_sched_barrier(0);
      pd(std::integral_constant<int, 4>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (IL && ST == NST - 1 && J < ULD) {
      __builtin_amdgcn_sched_barrier(0);
      wd(j_c);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using std::integral_constant;
  dma(integral_constant<int, 0>{});
  acc[X][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][0], b[S][1], acc[X][0][1], 0, 0, 0);
  dma(integral_constant<int, 1>{});
  acc[X][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][1], b[S][0], acc[X][1][0], 0, 0, 0);
  if constexpr (ST == 1 || ST == 2) {
    __builtin_amdgcn_sched_barrier(0);
    wino_half_math<ST - 1>(h, rows, o);
    __builtin_amdgcn_sched_barrier(0);
  }
  dma(integral_constant<int, 2>{});
  acc[X][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][1], b[S][1], acc[X][1][1], 0, 0, 0);
  dma(integral_constant<int, 3>{});
  acc[X][2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][2], b[S][0], acc[X][2][0], 0, 0, 0);
  if constexpr (ST == 1 || ST == 2) {
    __builtin_amdgcn_sched_barrier(0);
    wino_half_store<VB ^ 1, ST - 1>(h, o);
    __builtin_amdgcn_sched_barrier(0);
  }
  acc[X][2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][2], b[S][1], acc[X][2][1], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (ST + 1 < NST) wino_mma_steps<U, VB, ST + 1, IL>(acc, a, b, abase, bbase, h, rows, pd, wd);
}

// Eight accumulator registers -> LDS by `ds_write_addtid_b32` (address = M0 + offset + 4 * lane: no address
// register, 2 issue cycles instead of ds_write_b32's 4).  M0 also carries the LDS base of the LDS-DMA
// instructions, which the compiler tracks: it is saved and restored inside the statement.
template <int OFF>
__device__ __forceinline__ void wino_store8_addtid(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                                   float a7, unsigned base) {
  static_assert(OFF >= 0 && OFF + 7 * 256 < 65536, "ds offset field is 16 bits");
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %9\n\t"
      "s_nop 0\n\t"  // (M0 write -> add-tid LDS instruction: one wait state, not inserted for inline assembly)
      "ds_write_addtid_b32 %1 offset:%10\n\t"
      "ds_write_addtid_b32 %2 offset:%10+256\n\t"
      "ds_write_addtid_b32 %3 offset:%10+512\n\t"
      "ds_write_addtid_b32 %4 offset:%10+768\n\t"
      "ds_write_addtid_b32 %5 offset:%10+1024\n\t"
      "ds_write_addtid_b32 %6 offset:%10+1280\n\t"
      "ds_write_addtid_b32 %7 offset:%10+1536\n\t"
      "ds_write_addtid_b32 %8 offset:%10+1792\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(base), "n"(OFF)
      : "memory");
}

// A wave's two 32 x 32 accumulator tiles -> exchange buffer BUF (byte BUF * 64 KB of the kernel's LDS, which starts at
// offset 0 or close to it), Es[xi = 2 wave + x][r][lane].  The address of ds_write_addtid is
// M0[15:0] + a 16-bit immediate: for the second buffer the 64 KB are split between the two.
template <int BUF>
__device__ __forceinline__ void wino_store_round(const f32x16 &c0, const f32x16 &c1, int wave, unsigned lds0) {
  constexpr int M0_PART = BUF * 8188, IMM_PART = BUF * (ESZ * 4 - 8188);
  static_assert(7 * 8192 + M0_PART < 65536 && IMM_PART + 31 * 256 < 65536, "address split of the second exchange buffer");
  const unsigned base = __builtin_amdgcn_readfirstlane(lds0) + (unsigned)(wave * (XW * 16 * 256) + M0_PART);
  wino_store8_addtid<IMM_PART>(c0[0], c0[1], c0[2], c0[3], c0[4], c0[5], c0[6], c0[7], base);
  wino_store8_addtid<IMM_PART + 8 * 256>(c0[8], c0[9], c0[10], c0[11], c0[12], c0[13], c0[14], c0[15], base);
  wino_store8_addtid<IMM_PART + 16 * 256>(c1[0], c1[1], c1[2], c1[3], c1[4], c1[5], c1[6], c1[7], base);
  wino_store8_addtid<IMM_PART + 24 * 256>(c1[8], c1[9], c1[10], c1[11], c1[12], c1[13], c1[14], c1[15], base);
}

template <bool RES, bool D2W>
__global__ __launch_bounds__(kThreads) void wino_conv3x3_kernel(
    const float *__restrict__ in, const float *__restrict__ upk, float *out, int cin, int cin_pad, int h,
    int w, int cout, int ho, int wo, int tiles_r, int tiles_c, int cblocks, WView vin, WView vout, WEpilogue ep) {
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
  // (one DMA instruction: the steady-state chunk places them one by one between its MFMAs)
  auto patch_piece = [&](int chunk, int buf, int j) {  // (elements past PSZ re-read element 0 into the buffer's slack)
    // (uniform base + 32-bit byte offset of the lane: `global_load_lds_dword v, s[..]`, no 64-bit address
    // arithmetic in front of the DMA -- every instruction here costs the matrix pipe its issue time)
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
  // a weight stage = KC channels x the wave's two GEMMs x 96 floats = 3 KB contiguous in the packed
  // weights: three 16-byte LDS-DMA instructions
  const float *uw = upk + ((size_t)cb * kWaves + wave) * cin_pad * XW * CO;  // wave-uniform (scalar registers)
  float *us_w = Us + wave * URING * USZ;
  auto weight_piece = [&](int chunk, auto j_c) {
    constexpr int j = decltype(j_c)::value;
    glb_bytes_t *src = (glb_bytes_t *)(uw + (size_t)chunk * USZ);  // (uniform)
    float *dst = us_w + (chunk % URING) * USZ;
    // (opaque lane offset: hoisted out of the loop as a 64-bit per-lane pointer it costs two registers
    // the matrix loop does not have -- a spill there reloads through vmcnt, i.e. waits for every DMA)
    unsigned lo = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lo));
    // (the instruction's immediate offset moves both ends: piece j of the stage, in memory and in LDS)
    __builtin_amdgcn_global_load_lds((glb_ptr_t *)(src + lo), (lds_ptr_t *)dst, 16, j * 1024, 0);
  };
  auto issue_weights = [&](int chunk, bool guard) {
    if (guard && chunk >= nchunk) return;
    static_assert(ULD == 3, "weight pieces are issued by name");
    weight_piece(chunk, std::integral_constant<int, 0>{});
    weight_piece(chunk, std::integral_constant<int, 1>{});
    weight_piece(chunk, std::integral_constant<int, 2>{});
  };

  // ---- input transform: V = Bt d B.  wave -> channel (wave & 3) of the stage, lane -> tile; waves 0-3 take
  // the first two rows of Bt d (V values xi = 0..7 of the pair), waves 4-7 the other two: 8 LDS reads, 8 + 8
  // adds, 8 LDS writes per thread and chunk, placed between the MFMAs of the chunk before (wino_mma_steps).
  // transform_full is the same arithmetic for a whole pair, in one piece: stage 0, by waves 0-3.
  const int wave4 = wave >> 2, tch = wave & 3;  // (uniform)
  const int tty = lane >> 5, ttx = lane & 31;
  const float *tp0 = kFlat ? Ps + tch * PR * PC + 2 * lane       // (flat: one tile row, tile = lane)
                           : Ps + (tch * PR + 2 * tty) * PC + 2 * ttx;  // the pair's 4 x 4 block in patch stage 0
  float *tv0 = Vs + tch * (TX * TY) + lane;                     // its slot in V buffer 0
  const WinoHalf half_t = {tp0, tv0 + wave4 * 8 * (KC * TX * TY), wave4 ? 2 * PC : 0, wave4 ? PC : 2 * PC,
                           wave4 ? 3 * PC : 2 * PC, wave4 ? -1.f : 1.f};
  auto transform_full = [&](int pbuf, int vbuf) {
    const float *p = tp0 + pbuf * PBUF;
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const f32x2 lo = *reinterpret_cast<const f32x2 *>(p + r * PC), hi = *reinterpret_cast<const f32x2 *>(p + r * PC + 2);
      d[r][0] = lo.x, d[r][1] = lo.y, d[r][2] = hi.x, d[r][3] = hi.y;
    }
    float *v = tv0 + vbuf * VSZ;
    constexpr int XS = KC * TX * TY;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float t0 = d[0][c] - d[2][c], t1 = d[1][c] + d[2][c], t2 = d[2][c] - d[1][c], t3 = d[1][c] - d[3][c];
      d[0][c] = t0, d[1][c] = t1, d[2][c] = t2, d[3][c] = t3;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[(i * 4 + 0) * XS] = d[i][0] - d[i][2];
      v[(i * 4 + 1) * XS] = d[i][1] + d[i][2];
      v[(i * 4 + 2) * XS] = d[i][2] - d[i][1];
      v[(i * 4 + 3) * XS] = d[i][1] - d[i][3];
    }
  };

  f32x16 acc[XW][3][2];
#pragma unroll
  for (int x = 0; x < XW; x++)
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[x][m][n][r] = 0.f;

  // ---- prologue ----
  // bias / PReLU slope of the cout block -> a table behind the stage memory, by LDS-DMA with the first
  // stages (waves 0-3: one 64-float piece each; couts past the end repeat the last one).  Loaded into
  // registers in front of the way out they were one more exposed memory round trip per tile.
  float *Bt = lds + kStageFloats;
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
  if (wave4 == 0) transform_full(0, 0);

  // per-lane LDS byte addresses of the operand fragments (see wino_read_step): A = weights
  // [stage][ci = 2 kp + half][x][96], B = V[vbuf][xi = 2 wave + x][ci = 2 kp + half][tile]
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);  // low half of the flat address = LDS offset
  if (lds0 != 0) __builtin_trap();  // (uniform; see the LDS contract in the header comment)
  const unsigned abase = lds0 + (unsigned)((us_w - lds) + half * XW * CO + l31) * 4u;
  const unsigned bbase = lds0 + (unsigned)((Vs - lds) + ((wave * XW) * KC + half) * (TX * TY) + l31) * 4u;

  // One chunk (KC input channels).  U = chunk % 4 is a compile-time ring slot (the loop below is unrolled
  // over four chunks), so are the patch / V buffers (U & 1).  DMA issue order of a wave, one patch stage
  // and one weight stage per chunk:  ... patch(chunk+1), weights(chunk+3) | patch(chunk+2), weights(chunk+4).
  // On arrival patch(chunk+1) -- and with it everything older, weights(chunk) included -- must have landed;
  // weights(chunk+3), issued a moment ago, stays in flight (counted wait).  Then everybody's: V(chunk) is
  // complete (lgkmcnt(0): this thread's LDS writes) and the MFMAs of chunk-1, last readers of V's other
  // buffer, are done.  Wait and barrier are one asm statement: __syncthreads() would wait for vmcnt(0).
  // STEADY = every stream is still issuing (chunk + 4 < nchunk, chunk > 0): no guards, counted wait.
#ifdef PCONV_WINO_STAMP
  unsigned long long st_bar = 0, st_head = 0, st_mm = 0, st_wd = 0, st_n = 0;
#endif
  auto body = [&](auto u_c, auto steady_c, int chunk) {
    constexpr int U = decltype(u_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    constexpr int vb = U & 1;
    WINO_STAMP(t0);
    if (STEADY)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(ULD) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    WINO_STAMP(t1);
    // steady state: the DMA instructions go inside the matrix block; head and tail chunks, where a stream may
    // have ended, keep them around it (a guard inside the block would be a join with 192 live accumulators)
    if (!STEADY) issue_patch(chunk + 2, vb, true);  // (that stage was read by the transform of this chunk, before the barrier)
    WINO_STAMP(t2);
    // the matrix block, with this wave's half of the transform patch(chunk+1) -> V(chunk+1) inside (behind
    // the last chunk it turns stale patch bytes into V values nobody reads)
    float a[2][3], bv[2][2];
    WinoRows rows;
    wino_read_step<U, vb, 0>(a[0], bv[0], abase, bbase);
    // (the last piece holds elements 1536 .. 1583 of the stage: wave 0's lanes 0-47; the other waves would only
    // re-read element 0 into the slack.  The counted wait at the barrier counts the YOUNGEST instructions, the
    // weight pieces, so the waves need not issue the same number of older ones.)
    auto pd = [&](auto j_c) {
      constexpr int J = decltype(j_c)::value;
      if (J * kThreads + 64 >= PSZ && wave != 0) return;
      patch_piece(chunk + 2, vb, J);
    };
    auto wd = [&](auto j_c) { weight_piece(chunk + URING, j_c); };
    wino_mma_steps<U, vb, 0, STEADY>(acc, a, bv, abase, bbase, half_t, rows, pd, wd);
    WINO_STAMP(t3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (this wave's reads of the weight stage are complete -- their values fed the MFMAs: refill it)
    if (!STEADY) issue_weights(chunk + URING, true);
#ifdef PCONV_WINO_STAMP
    WINO_STAMP(t4);
    if (STEADY) st_bar += t1 - t0, st_head += t2 - t1, st_mm += t3 - t2, st_wd += t4 - t3, st_n += 1;
#endif
  };
  using std::integral_constant;
  auto group = [&](auto steady_c, int c4) {
    body(integral_constant<int, 0>{}, steady_c, c4);
    body(integral_constant<int, 1>{}, steady_c, c4 + 1);
    body(integral_constant<int, 2>{}, steady_c, c4 + 2);
    body(integral_constant<int, 3>{}, steady_c, c4 + 3);
  };
  // nchunk is a multiple of 4 (cin % 16 == 0).  Three plain loops -- head, steady state, tail -- rather
  // than one loop choosing between two bodies: with both versions inside one loop the register allocator
  // moved accumulator tiles through scratch memory at the joins (5 x slower: scratch reloads wait on vmcnt,
  // i.e. on every DMA in flight).
  int c4 = 0;
  group(integral_constant<bool, false>{}, c4);
  c4 += 4;
#pragma unroll 1
  for (; c4 + 8 <= nchunk; c4 += 4) group(integral_constant<bool, true>{}, c4);
#pragma unroll 1
  for (; c4 < nchunk; c4 += 4) group(integral_constant<bool, false>{}, c4);
  const int erow = wave * 2 + half, ecol = l31;  // cout inside the block (second pair: + 16; d2w: cout pair), tile column
#ifdef PCONV_WINO_STAMP
  if (blockIdx.x == gridDim.x / 2 + 1 && lane == 0) {
    unsigned long long *o = wino_stamps[wave];
    o[0] = st_bar, o[1] = st_head, o[2] = st_mm, o[3] = st_wd, o[4] = 0, o[5] = st_n;
  }
#endif
  __syncthreads();  // all MFMAs done: the stage memory becomes the exchange buffer

  // ---- output transform + epilogue ----
  // One 32-cout x 32-tile block of all sixteen M[xi] per round, through one of two exchange buffers
  // (a round's readers are past their reads when they arrive at the next round's barrier: one barrier
  // per round).  A thread finishes two (cout, tile) pairs per round (d2w: one pair of couts of a tile).
  // What the way out reads from memory (the residual) is requested before the exchange.
  const int act = ep.act;
  const int trim_at = ep.trim ? limit : wo;
  const float *resp = RES ? ep.residual + (size_t)t * ep.vres.ts : nullptr;
  // accumulator half n of a wave = tile row n (4-row workgroup) or tile columns 32 n .. 32 n + 31 (flat)
  auto ocol_of = [&](int n) { return kFlat ? c0 + 2 * (32 * n + ecol) : c0 + 2 * ecol; };
  auto orow_of = [&](int n) { return kFlat ? r0 : r0 + 2 * n; };
  // residual values of round k = (m, n): requested two rounds ahead (a round is ~1 us, a load from HBM
  // under load 2-3 us: requested at the head of their own round every one of the six waits was exposed)
  struct ResPair {
    f32x2 v[2][2];
  };
  // (addresses clamped into the view instead of guarded, and the kind of layer -- residual / plain / Dtow
  // -- a template parameter of the kernel instead of a uniform branch: with a branch around a load the
  // compiler waits as if the loads behind it had not been issued, i.e. for vmcnt(0..3): for the residual
  // of two rounds ahead and for the stores of the round before)
  auto load_res = [&](int round) {
    ResPair rp = {{{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}}};
    if (RES && round < 6) {
      const int m = round >> 1, n = round & 1;
      const int oc = ocol_of(n) < wo ? ocol_of(n) : wo - 2;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        int co = cout0 + m * 32 + erow + 16 * j;
        co = co < cout ? co : cout - 1;
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++) {
          int rr = orow_of(n) + a2;
          rr = rr < ho ? rr : ho - 1;
          rp.v[j][a2] = *reinterpret_cast<const f32x2 *>(resp + (size_t)co * ep.vres.cs + (size_t)rr * ep.vres.rs + oc);
        }
      }
    }
    return rp;
  };
  ResPair rq[2] = {load_res(0), load_res(1)};
#pragma unroll
  for (int m = 0; m < 3; m++) {
#pragma unroll
    for (int n = 0; n < 2; n++) {
      float *Es = lds + ((m * 2 + n) & 1) * ESZ;  // [xi][32 couts][32 tiles]
      const int orow = orow_of(n), ocol = ocol_of(n);
      const ResPair rcur = rq[(m * 2 + n) & 1];
      rq[(m * 2 + n) & 1] = load_res(m * 2 + n + 2);
      // Es[xi][r][lane]: accumulator register r of lane (half, l31) = cout row (r & 3) + 8 (r >> 2) + 4 half
      if (((m * 2 + n) & 1) == 0)
        wino_store_round<0>(acc[0][m][n], acc[1][m][n], wave, lds0);
      else
        wino_store_round<1>(acc[0][m][n], acc[1][m][n], wave, lds0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (stores issued by hand: the compiler does not count them)
      __syncthreads();
      if (!D2W) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int row = erow + 16 * j;
          float mm[16];
#pragma unroll
          for (int xi = 0; xi < 16; xi++) mm[xi] = Es[xi * 1024 + ((row & 3) + 4 * (row >> 3)) * 64 + ((row >> 2) & 1) * 32 + ecol];
          const int co = cout0 + m * 32 + row;
          if (co < cout && ocol < wo) {
            const float bco = Bt[m * 32 + row], sl = Bt[128 + m * 32 + row];
            // Y = At M A, M[i][j] = mm[4 i + j]
            float ta[2][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              ta[0][q] = mm[q] + mm[4 + q] + mm[8 + q];
              ta[1][q] = mm[4 + q] - mm[8 + q] - mm[12 + q];
            }
#pragma unroll
            for (int a2 = 0; a2 < 2; a2++) {
              if (orow + a2 >= ho) continue;
              float y0 = ta[a2][0] + ta[a2][1] + ta[a2][2] + bco;
              float y1 = ta[a2][1] - ta[a2][2] - ta[a2][3] + bco;
              if (act == 1) {
                y0 = y0 < 0 ? y0 * sl : y0;
                y1 = y1 < 0 ? y1 * sl : y1;
              }
              if (RES) {
                y0 = rcur.v[j][a2].x + y0;
                y1 = rcur.v[j][a2].y + y1;
              }
              if (ocol >= trim_at) y0 = 0.f;
              if (ocol + 1 >= trim_at) y1 = 0.f;
              float *q = outp + (size_t)co * vout.cs + (size_t)(orow + a2) * vout.rs + ocol;
              f32x2 yv = {y0, y1};
              *reinterpret_cast<f32x2 *>(q) = yv;  // (wo is even: a 2x2 block never straddles the edge)
            }
          }
        }
      } else {
        // Dtow by the store: couts co (even, sx = 0) and co + 1 (sx = 1) of a tile -> 4 consecutive
        // outputs in each of 2 rows of channel co >> 2
        float m0[16], m1[16];
#pragma unroll
        for (int xi = 0; xi < 16; xi++) {
          // rows 2 erow and 2 erow + 1: same (row >> 2), register index differs by one
          const int ra = 2 * erow;
          m0[xi] = Es[xi * 1024 + ((ra & 3) + 4 * (ra >> 3)) * 64 + ((ra >> 2) & 1) * 32 + ecol];
          m1[xi] = Es[xi * 1024 + ((ra & 3) + 1 + 4 * (ra >> 3)) * 64 + ((ra >> 2) & 1) * 32 + ecol];
        }
        const int co = cout0 + m * 32 + 2 * erow;
        if (co < cout && ocol < wo) {
          const float b0 = Bt[m * 32 + 2 * erow], b1 = Bt[m * 32 + 2 * erow + 1];
          const float s0 = Bt[128 + m * 32 + 2 * erow], s1 = Bt[128 + m * 32 + 2 * erow + 1];
          const int cq = co >> 2, sy = (co >> 1) & 1;
#pragma unroll
          for (int a2 = 0; a2 < 2; a2++) {
            if (orow + a2 >= ho) continue;
            float t0[4], t1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              t0[q] = a2 == 0 ? m0[q] + m0[4 + q] + m0[8 + q] : m0[4 + q] - m0[8 + q] - m0[12 + q];
              t1[q] = a2 == 0 ? m1[q] + m1[4 + q] + m1[8 + q] : m1[4 + q] - m1[8 + q] - m1[12 + q];
            }
            float y00 = t0[0] + t0[1] + t0[2] + b0, y01 = t0[1] - t0[2] - t0[3] + b0;
            float y10 = t1[0] + t1[1] + t1[2] + b1, y11 = t1[1] - t1[2] - t1[3] + b1;
            if (act == 1) {
              y00 = y00 < 0 ? y00 * s0 : y00;
              y01 = y01 < 0 ? y01 * s0 : y01;
              y10 = y10 < 0 ? y10 * s1 : y10;
              y11 = y11 < 0 ? y11 * s1 : y11;
            }
            float *q = outp + (size_t)cq * vout.cs + (size_t)(2 * (orow + a2) + sy) * vout.rs + 2 * ocol;
            *reinterpret_cast<float4 *>(q) = make_float4(y00, y10, y01, y11);
          }
        }
      }
    }
  }
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

#ifndef PCONV_WINO_FLAT
#ifdef PCONV_WINO_STAMP
extern "C" int pconv_wino_read_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(wino_stamps), sizeof(wino_stamps)) == hipSuccess ? 0 : 1;
}
#endif

// floats of the packed Winograd weights of a (cout, cin, 3, 3) layer
extern "C" long long pconv_wino_packed_size(int cout, int cin) {
  const int cblocks = (cout + CO - 1) / CO, cin_pad = (cin + KC - 1) / KC * KC;
  return (long long)cblocks * 16 * cin_pad * CO;
}

extern "C" int pconv_wino_pack_weight(const float *w, float *packed, int cout, int cin, void *stream) {
  PCONV_REQUIRE(w && packed && cout > 0 && cin > 0, "wino_pack: bad argument");
  const int cblocks = (cout + CO - 1) / CO, cin_pad = (cin + KC - 1) / KC * KC;
  const long long total = (long long)cblocks * cin_pad * CO;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w, packed,
                     cout, cin, cin_pad, cblocks);
  PCONV_LAUNCH_CHECK("wino_pack_weight");
  return PCONV_OK;
}

// 1 when pconv_conv3x3_wino takes the layer (3x3 stride 1, even output size)
extern "C" int pconv_wino_supported(int cin, int h, int w, int cout, int d2w) {
  if (h < 4 || w < 4 || ((h - 2) & 1) || ((w - 2) & 1)) return 0;
  if (cin < 16 || cin % 16 || cout < 32) return 0;  // the 3-channel input layer and the 12-channel output layer stay direct
  if (d2w && (cout & 3)) return 0;
  return 1;
}

#define PCONV_WINO_ENTRY pconv_conv3x3_wino
#else   // the flat build exports its convolution only; weights are packed by the 4-row build (same layout)
#define PCONV_WINO_ENTRY pconv_conv3x3_wino_flat
#endif

extern "C" int PCONV_WINO_ENTRY(const float *in, const float *packed_u, const float *bias, float *out, int tn, int cin,
                                int h, int w, int cout, int act, const float *slope, const int32_t *col_limit,
                                int npart, const float *residual, int trim, int d2w, const long long *views,
                                void *stream) {
  PCONV_REQUIRE(in && packed_u && out, "conv3x3_wino: null pointer");
  PCONV_REQUIRE(pconv_wino_supported(cin, h, w, cout, d2w), "conv3x3_wino: unsupported shape %d x %d x %d -> %d", cin, h, w,
                cout);
  PCONV_REQUIRE(act == 0 || (act == 1 && slope), "conv3x3_wino: bad activation %d", act);
  PCONV_REQUIRE(!d2w || (!residual && !trim), "conv3x3_wino: depth-to-width takes no residual / trim");
  PCONV_REQUIRE(!col_limit || npart > 0, "conv3x3_wino: col_limit needs npart");
  PCONV_REQUIRE(!trim || col_limit, "conv3x3_wino: trim needs col_limit");
  PCONV_REQUIRE(residual != out, "conv3x3_wino: residual must not alias the output");
  const int ho = h - 2, wo = w - 2;
  const int oc = d2w ? cout / 4 : cout, oh = d2w ? 2 * ho : ho, ow = d2w ? 2 * wo : wo;
  const WView vin = view_at(views, 0, cin, h, w), vout = view_at(views, 1, oc, oh, ow);
  const WEpilogue ep = {bias, slope, residual, col_limit, npart, act, trim, d2w, view_at(views, 2, cout, ho, wo)};
  PCONV_REQUIRE(view_ok(vin, cin, h, w) && view_ok(vout, oc, oh, ow) && (!residual || view_ok(ep.vres, cout, ho, wo)),
                "conv3x3_wino: strides overlap");
  PCONV_REQUIRE(((long long)(KC - 1) * vin.cs + (long long)(h - 1) * vin.rs + w) * 4 < (1LL << 32),
                "conv3x3_wino: input channel stride too large for 32-bit byte offsets inside a chunk");
  // float2 / float4 accesses: rows of every view start on even element offsets
  PCONV_REQUIRE(vout.rs % 2 == 0 && vout.cs % 2 == 0 && vout.ts % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
                "conv3x3_wino: output rows must be 8-byte aligned");
  PCONV_REQUIRE(!residual || (ep.vres.rs % 2 == 0 && ep.vres.cs % 2 == 0 && ep.vres.ts % 2 == 0 &&
                              (reinterpret_cast<uintptr_t>(residual) & 7) == 0),
                "conv3x3_wino: residual rows must be 8-byte aligned");
  const int tiles_r = (ho + OROWS - 1) / OROWS, tiles_c = (wo + OCOLS - 1) / OCOLS;
  const int cblocks = (cout + CO - 1) / CO, cin_pad = (cin + KC - 1) / KC * KC;
  const long long grid = (long long)tn * tiles_r * tiles_c * cblocks;
  PCONV_REQUIRE(grid > 0 && grid <= 0x7fffffffLL, "conv3x3_wino: grid %lld out of range", grid);
  const size_t smem = (size_t)kLdsFloats * sizeof(float);
  // one instantiation per kind of layer: 0 plain, 1 residual, 2 depth-to-width
  using kernel_t = decltype(&wino_conv3x3_kernel<false, false>);
  static const kernel_t kernels[3] = {wino_conv3x3_kernel<false, false>, wino_conv3x3_kernel<true, false>,
                                      wino_conv3x3_kernel<false, true>};
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
        pconv_set_error("conv3x3_wino: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e));
        return PCONV_ELAUNCH;
      }
      raised[kind].fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kernels[kind], dim3((unsigned)grid), dim3(kThreads), smem, as_stream(stream), in, packed_u, out, cin,
                     cin_pad, h, w, cout, ho, wo, tiles_r, tiles_c, cblocks, vin, vout, ep);
  PCONV_LAUNCH_CHECK("conv3x3_wino");
  return PCONV_OK;
}
