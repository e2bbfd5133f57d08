// Dense per-tile convolution as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32).  This is the FLOP hot spot of the codec: the
// nn.Conv2d call sites of model_zoo_v2.py:41-45,83-86,100-105,119,143,158-164,
// 181,205 (cuDNN in the reference).
//
//   D[cout][pixel] = sum_k Wp[k][cout] * X[k][pixel],  k = (ci*KS + kh)*KS + kw
//
// Orientation: couts are the MFMA "row" operand (A), pixels the "column" operand
// (B), so an accumulator register holds 32 consecutive pixels of one output
// channel and every store instruction writes two 128-byte row segments of the
// NCHW output.  A workgroup (WM x WN waves) owns BM couts x ROWS output rows x 64
// columns.  Per chunk of KC input channels the input patch and the weight slab
// go to LDS by LDS-DMA (global_load_lds, double buffered: the next chunk's DMA is
// issued before the MFMA block and has all of it to land).  Each wave keeps
// MT x NT 32x32 accumulators; per k-pair it reads MT + NT operands from LDS
// (conflict-free: 32 consecutive couts / pixels per lane group), one k-pair
// ahead of the MT*NT MFMAs of 64 cycles each that consume them.
//
// Measured (MI355X, 192->192 3x3, 16 x 64 x 2048 px): 127 TFLOP/s executed = 81 %
// of the 157.3 TFLOP/s fp32-MFMA peak; the same loop without staging runs 141,
// without staging and LDS reads 146.  Dead latitude columns are skipped on top
// (col_limit): -16 % time at half resolution.
//
// Numerics contract: every output is ONE k-ascending fmaf chain starting at 0
// (the MFMA accumulates k0 then k1 into the same register, chunks continue the
// chain), then + bias, then the activation.  The oracle restates exactly that.
#include <atomic>
#include <utility>
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileCols = 64;
// rows a batch of the pipelined way out requests ahead (both: the layer reads two tensors)
#ifndef PCONV_PIPE_ROWS
#define PCONV_PIPE_ROWS(both) ((both) ? 8 : 8)
#endif
#ifndef PCONV_KC1
#define PCONV_KC1 16  // input channels per LDS stage of the 1x1 layers
#endif
// quads requested ahead of their turn in conv_epilogue_quads (4 registers each; 8 for the GDN with a residual).
// Measured 2 / 4 / 6 / 8 ahead (profiles/round4_quad_ahead.txt): level within 2 %, two is never behind.
#ifndef PCONV_QUAD_AHEAD
#define PCONV_QUAD_AHEAD 2
#endif
#ifndef PCONV_QUAD_AHEAD_GDN
#define PCONV_QUAD_AHEAD_GDN 2
#endif

template <int ROWS, int KS, int S>
struct Patch {
  // sampling step of the staged patch (a strided 1x1 only needs every S-th pixel)
  static constexpr int PS = (KS == 1) ? S : 1;
  static constexpr int Q = S / PS;  // LDS step between neighbouring output pixels
  static constexpr int PR = (ROWS - 1) * Q + KS;
  static constexpr int PC = (kTileCols - 1) * Q + KS;
};

// WM x WN waves; a wave owns MT x NT accumulators of 32 couts x 32 pixels.  The
// workgroup's pixel tile is WN*NT segments of 32 columns = ROWS rows x 64 columns.
template <int MT, int NT, int WM, int WN, int KS, int S, int KC>
struct ConvCfg {
  static constexpr int THREADS = 64 * WM * WN;
  static constexpr int ROWS = WN * NT / 2;
  static constexpr int BM = 32 * MT * WM;
  static constexpr int KK = KC * KS * KS;  // reduction entries per chunk (even)
  using P = Patch<ROWS, KS, S>;
  static constexpr int XSZ = KC * P::PR * P::PC;
  static constexpr int WSZ = KK * BM;
  static constexpr int STAGE = XSZ + WSZ;
  static constexpr int XLD = (XSZ + THREADS - 1) / THREADS;        // floats / thread
  static constexpr int WLD = (WSZ / 4 + THREADS - 1) / THREADS;    // float4 / thread
  static constexpr int MTv = MT, NTv = NT, KSv = KS;
  // patch offset (floats) of reduction entry k = (ci*KS + kh)*KS + kw inside a chunk
  static constexpr int patch_off(int k) {
    return (k / (KS * KS)) * P::PR * P::PC + ((k / KS) % KS) * P::PC + (k % KS);
  }
  // the two lane halves of an MFMA take entries 2kp and 2kp+1: their patch offsets differ by one
  // of ND values (next column / next row / next channel), one LDS base register each
  static constexpr int ND = (KS == 1) ? 1 : 3;
  static constexpr int delta(int i) {
    return KS == 1 ? P::PR * P::PC : (i == 0 ? 1 : (i == 1 ? P::PC - (KS - 1) : P::PR * P::PC - (KS - 1) * P::PC - (KS - 1)));
  }
  static constexpr int delta_index(int kp) {
    const int d = patch_off(2 * kp + 1) - patch_off(2 * kp);
    return d == delta(0) ? 0 : (d == delta(1) ? 1 : 2);
  }
  static_assert((WN * NT) % 2 == 0, "pixel tile is whole rows of two 32-column segments");
  static_assert(KK % 2 == 0, "chunk reduction length must be even");
  static_assert(BM % 4 == 0, "weight slab rows are float4 multiples");
};

// What happens to an accumulator on its way out:  v = acc + bias;
//   act 1: PReLU(slope per cout)   act 4: sigmoid
//   act 2 / 3: GDN / inverse GDN (with SQ: the 1x1 "convolution" ran on the squared
//              input): v = x / sqrt(v) or x * sqrt(v), x = the input at the same place
//   gate:      v = gate[.] * v        (attention: trunk * sigmoid(conv))
//   residual:  v = residual[.] + v    (the "x + f(x)" of the residual blocks)
//   trim:      v = 0 from the tile's col_limit on (PseudoFill of the block output)
//   d2w:       the pixel shuffle of Dtow (dtow_cuda.cu:38-75) applied by the store: cout
//              4c'+2sy+sx at (row, col) goes to channel c' at (2 row + sy, 2 col + sx); a
//              lane holds the sx = 0/1 pair in neighbouring registers and stores it as one
//              float2, so the wave still writes whole 256-byte runs
// gate and residual have the output's shape.  This replaces up to four
// element-wise passes over the activation that follow the convolution in the
// reference's graph (sigmoid, mul, add, fill).
// element strides of a (tile, channel, row, column) tensor whose columns are contiguous:
// lets a convolution read from / write into the interior of a padded buffer
struct ConvView {
  long long ts, cs;
  int rs;
};

struct ConvEpilogue {
  const float *bias, *slope, *residual, *gate;
  const int32_t *col_limit;  // per latitude tile: first dead output column (may be null)
  int npart, act, trim;
  ConvView vres, vgate;
  int d2w;  // depth-to-width x2 on the way out (DtowOp fused): out is (cout/4, 2*ho, 2*wo)
};


// a workgroup tile that lies entirely in dead columns: zeros (with the Dtow shuffle when fused)
template <int BM, int ROWS, int THREADS>
__device__ __forceinline__ void conv_zero_tile(float *outp, const ConvView &vout, int d2w, int cout0, int cout,
                                               int r0, int c0, int ho, int wo, int tid) {
  for (int e = tid; e < BM * ROWS * kTileCols; e += THREADS) {
    const int col = e % kTileCols, row = (e / kTileCols) % ROWS, co = e / (kTileCols * ROWS);
    if (cout0 + co < cout && r0 + row < ho && c0 + col < wo) {
      const int cg = cout0 + co;
      if (d2w)
        outp[(size_t)(cg >> 2) * vout.cs + (size_t)(2 * (r0 + row) + ((cg >> 1) & 1)) * vout.rs +
             2 * (c0 + col) + (cg & 1)] = 0.f;
      else
        outp[(size_t)cg * vout.cs + (size_t)(r0 + row) * vout.rs + c0 + col] = 0.f;
    }
  }
}

// The way out of the accumulators (see ConvEpilogue), shared by the tiled and the
// weight-resident kernels.  Wave (wm, wn) holds MT x NT tiles of 32 couts x 32 pixels:
// reg r of a tile = cout row (r&3) + 8*(r>>2) + 4*half, pixel column = l31; pixel
// segment seg = wn*NT + n is row seg/2, 32-column half seg%2 of the workgroup tile.
template <int MT, int NT, int WN, int EB = 16>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MT][NT], const ConvEpilogue &ep, const float *inp,
                                              float *outp, const ConvView &vin, const ConvView &vout, int t, int r0,
                                              int c0, int cout0, int cout, int ho, int wo, int wm, int wn, int l31,
                                              int half, const float *bias_s, const float *slope_s) {
  // bias_s / slope_s: the cout block's bias and PReLU slope in LDS (zeros where the layer has none), indexed
  // by the cout inside the block.  Loaded from memory row by row, in front of their use, each load sat behind
  // the store of the row before and each store behind that load (vmcnt counts both, and the compiler waits
  // for vmcnt(0) around every guarded load): 2 x 16 x MT serial round trips per lane, ~75 of the ~95 us a
  // 1x1 workgroup lived.
  const int32_t *__restrict__ col_limit = ep.col_limit;
  const int npart = ep.npart, act = ep.act;
  const int trim_at = ((ep.trim || act == 2 || act == 3) && col_limit) ? col_limit[t % npart] : wo;
  const float *resp = ep.residual ? ep.residual + (size_t)t * ep.vres.ts : nullptr;
  const float *gatep = ep.gate ? ep.gate + (size_t)t * ep.vgate.ts : nullptr;
  // epilogue (see ConvEpilogue).  reg r of a 32x32 tile: cout row
  // (r&3) + 8*(r>>2) + 4*half, pixel column = l31.
  if (ep.d2w) {
#pragma unroll
    for (int m = 0; m < MT; m++) {
#pragma unroll
      for (int rp = 0; rp < 8; rp++) {
        const int r = 2 * rp;  // registers r, r+1: couts co (even), co+1 = sx 0, 1
        const int co = cout0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= cout) continue;
        const float b0 = bias_s[co - cout0], b1 = bias_s[co - cout0 + 1];
        const float s0 = slope_s[co - cout0], s1 = slope_s[co - cout0 + 1];
        const int cq = co >> 2, sy = (co >> 1) & 1;
#pragma unroll
        for (int n = 0; n < NT; n++) {
          const int seg = wn * NT + n;
          const int orow = r0 + (seg >> 1), ocol = c0 + (seg & 1) * 32 + l31;
          if (orow < ho && ocol < wo) {
            float2 v = make_float2(acc[m][n][r] + b0, acc[m][n][r + 1] + b1);
            if (act == 1) {
              if (v.x < 0) v.x = v.x * s0;
              if (v.y < 0) v.y = v.y * s1;
            }
            *reinterpret_cast<float2 *>(outp + (size_t)cq * vout.cs + (size_t)(2 * orow + sy) * vout.rs + 2 * ocol) = v;
          }
        }
      }
    }
    return;
  }
  static_assert(16 % EB == 0, "epilogue batches are whole fractions of a 32-cout tile");
#pragma unroll
  for (int m = 0; m < MT; m++) {
    // the residual values of EB of the 16 cout rows a lane holds of a 32-cout tile are requested first,
    // in one batch: one by one in front of their use (the compiler cannot move a load above the previous
    // store of `out`) every one of the 16 x NT round trips was exposed -- 8 % of the kernel on the
    // residual layers.  EB = 16: the whole tile at once (most in flight); the 1x1 kernels built for
    // three workgroups per CU take EB = 8 (16 registers fewer).
#pragma unroll
    for (int rb = 0; rb < 16; rb += EB) {
      float rv[EB][NT], q1[EB][NT];  // q1: the GDN's own input, or the attention gate (never both)
      const bool gdn = act == 2 || act == 3;
      auto batch = [&](float (&dst)[EB][NT], const float *src, const ConvView &v) {
#pragma unroll
        for (int rr = 0; rr < EB; rr++) {
          const int r = rb + rr;
          const int co = cout0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
          for (int n = 0; n < NT; n++) {
            const int seg = wn * NT + n;
            const int orow = r0 + (seg >> 1), ocol = c0 + (seg & 1) * 32 + l31;
            dst[rr][n] = (co < cout && orow < ho && ocol < wo) ? src[(size_t)co * v.cs + (size_t)orow * v.rs + ocol] : 1.f;
          }
        }
      };
      if (resp) batch(rv, resp, ep.vres);
      if (gdn) batch(q1, inp, vin);  // (likewise everything else the way out reads)
      if (gatep) batch(q1, gatep, ep.vgate);
#pragma unroll
      for (int rr = 0; rr < EB; rr++) {
        const int r = rb + rr;
        const int co = cout0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= cout) continue;
        const float bco = bias_s[co - cout0], sl = slope_s[co - cout0];
#pragma unroll
        for (int n = 0; n < NT; n++) {
          const int seg = wn * NT + n;
          const int orow = r0 + (seg >> 1), ocol = c0 + (seg & 1) * 32 + l31;
          if (orow < ho && ocol < wo) {
            const size_t oi = (size_t)co * vout.cs + (size_t)orow * vout.rs + ocol;
            float v = acc[m][n][r] + bco;
            if (act == 1) {
              if (v < 0) v = v * sl;
            } else if (gdn) {
              // 1x1, stride 1: input and output share their geometry
              const float xv = q1[rr][n];
              const float nrm = sqrtf(v);
              v = act == 2 ? xv / nrm : xv * nrm;
            } else if (act == 4) {
              v = 1.f / (1.f + expf(-v));
            }
            if (gatep) v = q1[rr][n] * v;
            if (resp) v = rv[rr][n] + v;
            if (ocol >= trim_at) v = 0.f;
            outp[oi] = v;
          }
        }
      }
    }
  }
}

// The way out of the 1x1 / GDN layers without gate, sigmoid or depth-to-width store, as straight-line code: what a
// row reads from memory (RES: the residual, SQ: the GDN's own input) is requested a batch of EB rows ahead, with
// addresses clamped into the tensors instead of guards around the loads, so that the compiler counts loads and
// stores (vmcnt(n)) and a batch's wait leaves the next batch and the stores of the one before in flight.  In
// conv_epilogue a 1x1 workgroup spends 62 % of its life in "request 16 rows, wait for all of them, store 16 rows",
// three times (profiles/round3_1x1_timeline.txt).  Same operations per output, in the same order: identical bits.
template <int MT, int NT, int WN, bool SQ, bool RES>
__device__ __forceinline__ void conv_epilogue_pipe(f32x16 (&acc)[MT][NT], const ConvEpilogue &ep, const float *inp,
                                                   float *outp, const ConvView &vin, const ConvView &vout, int t, int r0,
                                                   int c0, int cout0, int cout, int ho, int wo, int wm, int wn, int l31,
                                                   int half, const float *bias_s, const float *slope_s) {
  const int act = ep.act;
  const int trim_at = ((ep.trim || act == 2 || act == 3) && ep.col_limit) ? ep.col_limit[t % ep.npart] : wo;
  const float *resp = RES ? ep.residual + (size_t)t * ep.vres.ts : nullptr;
  constexpr int NROW = MT * 16, EB = PCONV_PIPE_ROWS(SQ && RES), NB = NROW / EB;
  struct RowIn {
    float r[NT], x[NT];
  };
  auto row_cout = [&](int q) { return cout0 + (wm * MT + (q >> 4)) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * half; };
  auto request = [&](int q) {
    RowIn v;
    int co = row_cout(q);
    co = co < cout ? co : cout - 1;
#pragma unroll
    for (int n = 0; n < NT; n++) {
      const int seg = wn * NT + n;
      int orow = r0 + (seg >> 1), ocol = c0 + (seg & 1) * 32 + l31;
      orow = orow < ho ? orow : ho - 1;
      ocol = ocol < wo ? ocol : wo - 1;
      v.r[n] = RES ? resp[(size_t)co * ep.vres.cs + (size_t)orow * ep.vres.rs + ocol] : 0.f;
      v.x[n] = SQ ? inp[(size_t)co * vin.cs + (size_t)orow * vin.rs + ocol] : 0.f;  // 1x1, stride 1: same geometry
    }
    return v;
  };
  RowIn cur[EB], nxt[EB];
  if (RES || SQ) {
#pragma unroll
    for (int e = 0; e < EB; e++) cur[e] = request(e);
  }
#pragma unroll
  for (int bi = 0; bi < NB; bi++) {
    if ((RES || SQ) && bi + 1 < NB) {
#pragma unroll
      for (int e = 0; e < EB; e++) nxt[e] = request((bi + 1) * EB + e);
    }
#pragma unroll
    for (int e = 0; e < EB; e++) {
      const int q = bi * EB + e, m = q >> 4, r = q & 15;
      const int co = row_cout(q);
      if (co < cout) {
        const float bco = bias_s[co - cout0], sl = slope_s[co - cout0];
#pragma unroll
        for (int n = 0; n < NT; n++) {
          const int seg = wn * NT + n;
          const int orow = r0 + (seg >> 1), ocol = c0 + (seg & 1) * 32 + l31;
          if (orow < ho && ocol < wo) {
            float v = acc[m][n][r] + bco;
            if (SQ) {
              const float xv = cur[e].x[n];
              const float nrm = sqrtf(v);
              v = act == 2 ? xv / nrm : xv * nrm;
            } else if (act == 1) {
              if (v < 0) v = v * sl;
            }
            if (RES) v = cur[e].r[n] + v;
            if (ocol >= trim_at) v = 0.f;
            outp[(size_t)co * vout.cs + (size_t)orow * vout.rs + ocol] = v;
          }
        }
      }
    }
    if ((RES || SQ) && bi + 1 < NB) {
#pragma unroll
      for (int e = 0; e < EB; e++) cur[e] = nxt[e];
    }
  }
}

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;
typedef const __attribute__((address_space(1))) char glb_bytes_t;

// f(integral_constant<I>) for I = I0 .. N - 1, unrolled by construction
template <int I, int N, class F>
__device__ __forceinline__ void static_for_(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_<I + 1, N>(f);
  }
}

// 16-byte global accesses addressed as "uniform base (scalar registers) + the lane's 32-bit byte offset": no
// 64-bit vector address arithmetic (the offset is made opaque so that it is not folded into a hoisted 64-bit add)
__device__ __forceinline__ float4 ld4_base_off(const float *base, unsigned off) {
  asm volatile("" : "+v"(off));
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = *reinterpret_cast<const __attribute__((address_space(1))) v4f *>((glb_bytes_t *)base + off);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_base_off(float *base, unsigned off, const float4 &v) {
  asm volatile("" : "+v"(off));
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) char glb_char_t;
  const v4f q = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<__attribute__((address_space(1))) v4f *>((glb_char_t *)base + off) = q;
}

// The way out of a tile of whole columns and a whole cout block (its last rows may hang over the lower edge) of the
// 1x1 / GDN layers in quads.  The accumulators
// hold one pixel per lane (32 consecutive pixels of a cout row per half wave), so the element-wise way out is 48
// rows x (4-byte load, ~20 address / guard / arithmetic instructions, 4-byte store) per lane: ~1 000 instructions
// per wave and tile, issued while the CU's other workgroup streams MFMAs -- and a vector instruction gets an issue
// slot only now and then while those stream (measured in the streamed experiment, DESIGN 4: ~25 cycles apiece).  Here the
// tile goes through the (by now free) stage memory, 32 * WM cout rows at a time: every wave parks its m-th
// accumulator tile, and the workgroup's 512 lanes take the round out as 4 quads each -- one 16-byte LDS read, one
// 16-byte load of the residual / the GDN's own input (requested one round ahead), the same arithmetic per
// element in the same order, one 16-byte store: ~4x fewer instructions, identical bits.
template <int MT, int WM, int WN, bool SQ, bool RES>
__device__ __forceinline__ void conv_epilogue_quads(f32x16 (&acc)[MT][1], const ConvEpilogue &ep, const float *in_t,
                                                    float *out_t, const ConvView &vin, const ConvView &vout, int t,
                                                    int r0, int c0, int cout0, int ho, int wo, int wm, int wn, int l31,
                                                    int half, float *park, const float *bias_s, const float *slope_s,
                                                    int tid) {
  constexpr int PX = 32 * WN, QROW = PX / 4, RROWS = 32 * WM, NQR = RROWS * QROW / 512, RSTEP = 512 / QROW;
  static_assert(WM * WN == 8 && RROWS * QROW % 512 == 0 && 32 % RSTEP == 0, "eight waves, whole quads per lane and round");
  const int act = ep.act;
  const int trim_at = ((ep.trim || act == 2 || act == 3) && ep.col_limit) ? ep.col_limit[t % ep.npart] : wo;
  const bool trims = trim_at < c0 + kTileCols;  // (uniform)
  const int row0 = tid / QROW, px = (tid % QROW) * 4;
  // (a tile may hang over the LOWER edge: a quad lies in one row; rows past the edge read the last row and are not
  // stored)
  const int orow_raw = r0 + px / kTileCols, ocol = c0 + px % kTileCols;
  const bool row_ok = orow_raw < ho;
  const int orow = row_ok ? orow_raw : ho - 1;
  // quad j of round m: parked row j * RSTEP + row0 = cout block row K(m, j) + row0, K uniform
  auto krow = [](int m, int j) { return ((j * RSTEP) / 32 * MT + m) * 32 + (j * RSTEP) % 32; };
  const unsigned lb_out = (unsigned)(((long long)row0 * vout.cs + (long long)orow * vout.rs + ocol) * 4);
  const unsigned lb_res = RES ? (unsigned)(((long long)row0 * ep.vres.cs + (long long)orow * ep.vres.rs + ocol) * 4) : 0u;
  const unsigned lb_in = (unsigned)(((long long)row0 * vin.cs + (long long)orow * vin.rs + ocol) * 4);
  const float *res_t = RES ? ep.residual + (size_t)t * ep.vres.ts + (size_t)cout0 * ep.vres.cs : nullptr;
  const float *x_t = in_t + (size_t)cout0 * vin.cs;  // (1x1, stride 1: input and output share their geometry)
  float *o_t = out_t + (size_t)cout0 * vout.cs;
  struct QuadIn {
    float4 r, x;
  };
  // what a quad reads from memory is requested D quads ahead of its turn (across the rounds: the addresses do not
  // depend on what is parked) and waits in a register ring of D + 1 quads; the twelve quads of a tile are
  // straight-line code, a ring slot is a NAME (flat quad index % (D + 1)), never copied
  constexpr int NF = MT * NQR;                 // quads per lane and tile
  constexpr int D = (SQ && RES) ? PCONV_QUAD_AHEAD_GDN : PCONV_QUAD_AHEAD;
  QuadIn ring[D + 1];
  auto request = [&](auto fc) {
    constexpr int f = decltype(fc)::value, m = f / NQR, j = f % NQR;
    if (RES) ring[f % (D + 1)].r = ld4_base_off(res_t + (size_t)krow(m, j) * ep.vres.cs, lb_res);
    if (SQ) ring[f % (D + 1)].x = ld4_base_off(x_t + (size_t)krow(m, j) * vin.cs, lb_in);
  };
  auto finish = [&](float v, float xv, float rv, float sl, bool trimmed, auto preluc, auto trimc) {
    if (SQ) {
      const float nrm = sqrtf(v);
      v = act == 2 ? xv / nrm : xv * nrm;
    } else if (decltype(preluc)::value) {
      if (v < 0) v = v * sl;
    }
    if (RES) v = rv + v;
    if (decltype(trimc)::value && trimmed) v = 0.f;
    return v;
  };
  auto take_out = [&](auto fc, auto preluc, auto trimc) {
    constexpr int f = decltype(fc)::value, m = f / NQR, j = f % NQR;
    const QuadIn &src = ring[f % (D + 1)];
    const int cb = krow(m, j) + row0;  // cout inside the block
    const float4 o4 = *reinterpret_cast<const float4 *>(park + (j * RSTEP + row0) * PX + px);
    const float bco = bias_s[cb];
    const float sl = decltype(preluc)::value ? slope_s[cb] : 1.f;
    const float o[4] = {o4.x, o4.y, o4.z, o4.w};
    const float xs[4] = {src.x.x, src.x.y, src.x.z, src.x.w};
    const float rs[4] = {src.r.x, src.r.y, src.r.z, src.r.w};
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = finish(o[k] + bco, xs[k], rs[k], sl, ocol + k >= trim_at, preluc, trimc);
    if (row_ok) st4_base_off(o_t + (size_t)krow(m, j) * vout.cs, lb_out, make_float4(v[0], v[1], v[2], v[3]));
  };
  auto whole_tile = [&](auto preluc, auto trimc) {
    if constexpr (RES || SQ) static_for_<0, (D < NF ? D : NF)>([&](auto fc) { request(fc); });
    static_for_<0, MT>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      // park: reg r of a 32x32 tile = cout row (r&3) + 8*(r>>2) + 4*half, pixel l31 of segment wn
#pragma unroll
      for (int r = 0; r < 16; r++) park[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PX + wn * 32 + l31] = acc[m][0][r];
      // (bare barriers: __syncthreads() would wait for the loads in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      static_for_<0, NQR>([&](auto jc) {
        constexpr int f = m * NQR + decltype(jc)::value;
        if constexpr ((RES || SQ) && f + D < NF) request(std::integral_constant<int, f + D>{});
        take_out(std::integral_constant<int, f>{}, preluc, trimc);
      });
      if constexpr (m + 1 < MT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the round has been read)
    });
  };
  // (PReLU / trim as constants: a tile without PReLU whose columns are all alive, the common case, pays for neither)
  if (!SQ && act == 1) {
    if (trims)
      whole_tile(std::true_type{}, std::true_type{});
    else
      whole_tile(std::true_type{}, std::false_type{});
  } else {
    if (trims)
      whole_tile(std::false_type{}, std::true_type{});
    else
      whole_tile(std::false_type{}, std::false_type{});
  }
}

// The same for the depth-to-width store (Dtow fused: cout 4c' + 2sy + sx at (row, col) -> channel c' at
// (2 row + sy, 2 col + sx); bias / PReLU only): four consecutive output floats are two neighbouring pixels of two
// neighbouring couts, i.e. two 8-byte reads from neighbouring rows of the parked round, interleaved; the 64 lanes of
// a wave then write 1 KB of one output row.
template <int MT, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_quads_d2w(f32x16 (&acc)[MT][1], const ConvEpilogue &ep, float *out_t,
                                                        const ConvView &vout, int r0, int c0, int cout0, int wm, int wn,
                                                        int l31, int half, float *park, const float *bias_s,
                                                        const float *slope_s, int tid) {
  constexpr int PX = 32 * WN, PROW = PX / 2, RROWS = 32 * WM, NQR = (RROWS / 2) * PROW / 512, PSTEP = 512 / PROW;
  static_assert(WM * WN == 8 && (RROWS / 2) * PROW % 512 == 0 && 16 % PSTEP == 0, "eight waves, whole quads per lane and round");
  const bool prelu = ep.act == 1;
  const int pair0 = tid / PROW, px = (tid % PROW) * 2;          // row pair inside the round, first pixel
  const int orow = r0 + px / kTileCols, ocol = c0 + px % kTileCols;
  // row pair j * PSTEP + pair0 of round m = couts K(m, j) + 2 pair0, + 1 of the block, K uniform and even
  auto krow = [](int m, int j) { return ((2 * j * PSTEP) / 32 * MT + m) * 32 + (2 * j * PSTEP) % 32; };
#pragma unroll
  for (int m = 0; m < MT; m++) {
#pragma unroll
    for (int r = 0; r < 16; r++) park[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PX + wn * 32 + l31] = acc[m][0][r];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int j = 0; j < NQR; j++) {
      const int cb = krow(m, j) + 2 * pair0;  // even cout inside the block: sx = 0; cb + 1: sx = 1
      const float2 a = *reinterpret_cast<const float2 *>(park + (2 * (j * PSTEP + pair0)) * PX + px);
      const float2 b = *reinterpret_cast<const float2 *>(park + (2 * (j * PSTEP + pair0) + 1) * PX + px);
      const float ba = bias_s[cb], bb = bias_s[cb + 1];
      float v[4] = {a.x + ba, b.x + bb, a.y + ba, b.y + bb};
      if (prelu) {
        const float sa = slope_s[cb], sb = slope_s[cb + 1];
        if (v[0] < 0) v[0] = v[0] * sa;
        if (v[1] < 0) v[1] = v[1] * sb;
        if (v[2] < 0) v[2] = v[2] * sa;
        if (v[3] < 0) v[3] = v[3] * sb;
      }
      const int co = cout0 + cb;
      float *dst = out_t + (size_t)(co >> 2) * vout.cs + (size_t)(2 * orow + ((co >> 1) & 1)) * vout.rs + 2 * ocol;
      *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (m + 1 < MT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// LDS-DMA of one chunk (input patch + weight slab): XLD patch dwords, then WLD weight float4s per
// thread, all issued at the head of the previous chunk's matrix loop.  (Issuing them one piece every
// second k-pair instead -- so that the ~30 KB do not come back in one burst -- measured 1.5 % slower.)
template <class C>
struct ConvStager {
  using P = typename C::P;
  static constexpr int PIECES = C::XLD + C::WLD;
  const float *xb, *wb;  // global bases of the chunk being staged
  float *xs, *ws;        // its LDS buffer
  const unsigned (&xoffs)[C::XLD];
  const unsigned (&woffs)[C::WLD];
  int tid, wave, tail;
  long long cs;
  bool active, ragged;

  template <int J>
  __device__ __forceinline__ void issue() const {
    if (!active) return;
    if constexpr (J < C::XLD) {
      const int e = tid + J * C::THREADS;
      if (e < C::XSZ) {
        unsigned off = xoffs[J];  // bytes
        if (ragged) {
          // channels past cin: read the last real one instead -- their rows of the
          // packed weight are zero, so the product adds +0 to the chain
          const int ci = e / (P::PC * P::PR);
          if (ci >= tail) off -= (unsigned)((ci - tail + 1) * cs * 4);
        }
        // (uniform base + 32-bit byte offset of the lane: `global_load_lds_dword v, s[..]`, no 64-bit vector
        // add in front of the DMA)
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_global_load_lds((glb_ptr_t *)((glb_bytes_t *)xb + off),
                                         (lds_ptr_t *)(xs + J * C::THREADS + wave * 64), 4, 0, 0);
      }
    } else {
      constexpr int JW = J - C::XLD;
      const int e4 = tid + JW * C::THREADS;
      if (e4 < C::WSZ / 4) {
        unsigned off = woffs[JW];  // bytes
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_global_load_lds((glb_ptr_t *)((glb_bytes_t *)wb + off),
                                         (lds_ptr_t *)(ws + (JW * C::THREADS + wave * 64) * 4), 16, 0, 0);
      }
    }
  }

  template <int J = 0>
  __device__ __forceinline__ void issue_all() const {
    if constexpr (J < PIECES) {
      issue<J>();
      issue_all<J + 1>();
    }
  }
};

// LDS operand reads of the matrix loop, issued by hand: `ds_read_b32 dst, base offset:imm` with the
// whole (chunk-invariant) offset in the immediate -- no address arithmetic in the loop -- and
// COUNTED waits: before the MFMAs of k-pair kp only the reads of kp must have landed, the MT + NT
// reads of kp+1 issued after them stay in flight (LDS returns in order).  The compiler's own
// scoreboard put `s_waitcnt lgkmcnt(0)` in front of every second MFMA block, i.e. waited for the
// reads it had just issued: their latency -- longer while the LDS-DMA of the next chunk is landing --
// was exposed once per 6 MFMAs.
template <int OFF>
__device__ __forceinline__ float lds_read_imm(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <class C, int KP>
__device__ __forceinline__ void conv_read_pair(float (&A)[C::MTv], float (&B)[C::NTv], unsigned abase,
                                               const unsigned (&bbase)[C::NTv][C::ND]) {
  static_assert(C::MTv <= 3 && C::NTv <= 2, "operand reads are written out for MT <= 3, NT <= 2");
  constexpr int aoff = 2 * KP * C::BM * 4;
  A[0] = lds_read_imm<aoff>(abase);
  if constexpr (C::MTv > 1) A[1] = lds_read_imm<aoff + 128>(abase);
  if constexpr (C::MTv > 2) A[2] = lds_read_imm<aoff + 256>(abase);
  constexpr int boff = C::patch_off(2 * KP) * 4, di = C::delta_index(KP);
  B[0] = lds_read_imm<boff>(bbase[0][di]);
  if constexpr (C::NTv > 1) B[1] = lds_read_imm<boff>(bbase[1][di]);
}

// wait until at most PENDING LDS reads are outstanding; the operands pass through the statement so
// that nothing that uses them can be scheduled above it
template <int PENDING, int MT, int NT>
__device__ __forceinline__ void conv_wait_pair(float (&A)[MT], float (&B)[NT]) {
  if constexpr (MT == 3 && NT == 1)
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(B[0]) : "n"(PENDING));
  else if constexpr (MT == 1 && NT == 1)
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(A[0]), "+v"(B[0]) : "n"(PENDING));
  else if constexpr (MT == 3 && NT == 2)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(B[0]), "+v"(B[1]) : "n"(PENDING));
  else
    static_assert(MT == 3 || MT == 1, "conv_wait_pair: unsupported tile");
}

#ifndef PCONV_PREFETCH
#define PCONV_PREFETCH 1
#endif
// operands are read PCONV_PREFETCH k-pairs ahead of the MFMAs that use them, through a ring of
// PCONV_PREFETCH + 1 register sets
constexpr int kAhead = PCONV_PREFETCH;

template <class C, bool SQ, int KP>
__device__ __forceinline__ void conv_chunk_step(f32x16 (&acc)[C::MTv][C::NTv], float (&a)[kAhead + 1][C::MTv],
                                                float (&b)[kAhead + 1][C::NTv], unsigned abase,
                                                const unsigned (&bbase)[C::NTv][C::ND]) {
  constexpr int NP = C::KK / 2, SETS = kAhead + 1;
  if constexpr (KP + kAhead < NP)
    conv_read_pair<C, KP + kAhead>(a[(KP + kAhead) % SETS], b[(KP + kAhead) % SETS], abase, bbase);
  constexpr int ahead = (NP - 1 - KP) < kAhead ? (NP - 1 - KP) : kAhead;  // k-pairs read after this one
  float(&A)[C::MTv] = a[KP % SETS];
  float(&B)[C::NTv] = b[KP % SETS];
  conv_wait_pair<ahead * (C::MTv + C::NTv)>(A, B);
  if (SQ) {
#pragma unroll
    for (int n = 0; n < C::NTv; n++) B[n] = B[n] * B[n];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < C::MTv; m++)
#pragma unroll
    for (int n = 0; n < C::NTv; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m], B[n], acc[m][n], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KP + 1 < NP) conv_chunk_step<C, SQ, KP + 1>(acc, a, b, abase, bbase);
}

template <class C, int KP>
__device__ __forceinline__ void conv_chunk_prologue(float (&a)[kAhead + 1][C::MTv], float (&b)[kAhead + 1][C::NTv],
                                                    unsigned abase, const unsigned (&bbase)[C::NTv][C::ND]) {
  if constexpr (KP < kAhead && KP < C::KK / 2) {
    conv_read_pair<C, KP>(a[KP], b[KP], abase, bbase);
    conv_chunk_prologue<C, KP + 1>(a, b, abase, bbase);
  }
}

// 1x1 layers (and the GDN contraction): waves per SIMD the kernel is compiled for and cout rows per
// epilogue batch (tuning knobs; the 3x3 kernels keep 4 waves per SIMD = two 8-wave workgroups per CU)
#ifndef PCONV_1X1_WAVES_EU
#define PCONV_1X1_WAVES_EU 2
#endif
#ifndef PCONV_1X1_EPI_ROWS
#define PCONV_1X1_EPI_ROWS 16
#endif

// cout rows per batch of the element-wise way out the ragged tiles of the quad kernels fall back to (few tiles:
// small batches keep the fallback out of the kernel's register count)
#ifndef PCONV_QUAD_FALLBACK_ROWS
#define PCONV_QUAD_FALLBACK_ROWS 4
#endif
#ifdef PCONV_CONV_STAMP
// profiling build: s_memtime of one workgroup per launch: [wave][entry, first chunk, end of the matrix loop, end,
// s_memrealtime at entry (100 MHz), SIMD/CU id]
__device__ unsigned long long conv_stamps[64][8][6];
#define CONV_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define CONV_STAMP(v)
#endif

// WAY: 0 = conv_epilogue (every epilogue, flags at run time), 1 / 2 = conv_epilogue_pipe without / with residual
template <int MT, int NT, int WM, int WN, int KS, int S, int KC, bool SQ, int WAY = 0>
__global__ __launch_bounds__(64 * WM * WN, (KS == 1 && S == 1 && MT == 3) ? PCONV_1X1_WAVES_EU : 2) void conv_mfma_kernel(
    const float *__restrict__ in, const float *__restrict__ wp, float *__restrict__ out, int cin, int h,
    int w, int cout, int cout_pad, int ho, int wo, int tiles_r, int tiles_c, int cblocks, int xcd_group, ConvView vin,
    ConvView vout, ConvEpilogue ep) {
  const int32_t *__restrict__ col_limit = ep.col_limit;
  const int npart = ep.npart;
  using C = ConvCfg<MT, NT, WM, WN, KS, S, KC>;
  using P = typename C::P;
  constexpr int kThreads = C::THREADS;
  constexpr int kTileRows = C::ROWS;
  extern __shared__ float lds[];
  CONV_STAMP(st0);

  // block -> (cout block, row tile, column tile, tile-batch index).  Row tiles run
  // fastest on purpose: workgroups are dealt to the 8 XCDs x 4 shader engines by
  // blockIdx % 32, and whether a workgroup is dead (col_limit) depends on its
  // COLUMN tile only -- with columns fastest and 32 column tiles per row the dead
  // ones all landed on the same engines and the live engines set the time
  // (measured: 50 % dead workgroups, 0 % less time).
  int b = blockIdx.x;
  if (xcd_group > 0) {
    // Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8, observed; used
    // for speed only).  Row-neighbour workgroups share input rows, so one column stripe
    // (xcd_group = cblocks * tiles_r consecutive logical blocks) is given to ONE XCD and
    // its shared rows are fetched into one L2 once; consecutive stripes rotate over the
    // XCDs, which also spreads the dead stripes (they depend on the column tile only).
    const int full = (int)(gridDim.x / (8u * xcd_group)) * 8 * xcd_group;
    if (b < full) {
      const int x = b & 7, j = b >> 3;
      b = (j / xcd_group) * 8 * xcd_group + x * xcd_group + j % xcd_group;
    }
  }
  const int cb = b % cblocks;
  b /= cblocks;
  const int trx = b % tiles_r;
  b /= tiles_r;
  const int tcx = b % tiles_c;
  const int t = b / tiles_c;
  const int r0 = trx * kTileRows, c0 = tcx * kTileCols;
  const int cout0 = cb * C::BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: LDS-DMA bases stay in scalar registers)
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  float *outp = out + (size_t)t * vout.ts;

  if (col_limit && c0 >= col_limit[t % npart]) {
    // tile lies entirely in the dead columns of this latitude band: zeros
    conv_zero_tile<C::BM, kTileRows, kThreads>(outp, vout, ep.d2w, cout0, cout, r0, c0, ho, wo, tid);
    return;
  }

  const float *inp = in + (size_t)t * vin.ts;
  const int nchunk = (cin + KC - 1) / KC;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

  // Staging is LDS-DMA (global_load_lds): each wave instruction moves 64 x 4 B
  // (patch) or 64 x 16 B (weight slab) from per-lane global addresses straight to
  // a lane-linear LDS range -- no staging registers, no ds_write pass, and no
  // address arithmetic in the chunk loop: the offsets are chunk-invariant up to a
  // uniform base and computed once here.  (Register staging with in-loop address
  // arithmetic cost 14 % of the kernel: the VALU shares its issue port with the
  // MFMAs.)
  unsigned xoffs[C::XLD], woffs[C::WLD];
#pragma unroll
  for (int j = 0; j < C::XLD; j++) {
    int e = tid + j * kThreads;
    e = e < C::XSZ ? e : 0;
    const int pc = e % P::PC;
    const int pr = (e / P::PC) % P::PR;
    const int ci = e / (P::PC * P::PR);
    int ir = r0 * S + pr * P::PS;
    int ic = c0 * S + pc * P::PS;
    ir = ir < h ? ir : h - 1;
    ic = ic < w ? ic : w - 1;
    xoffs[j] = (unsigned)((ci * vin.cs + (long long)ir * vin.rs + ic) * 4);  // bytes
  }
#pragma unroll
  for (int j = 0; j < C::WLD; j++) {
    int e4 = tid + j * kThreads;
    e4 = e4 < C::WSZ / 4 ? e4 : 0;
    const int kk = e4 / (C::BM / 4);
    const int co = (e4 % (C::BM / 4)) * 4;
    woffs[j] = (unsigned)(kk * cout_pad + co) * 4u;  // bytes
  }
  const int tail = cin % KC;  // channels of a ragged last chunk (0: none)
  const size_t xstep = (size_t)KC * vin.cs, wstep = (size_t)C::KK * cout_pad;

  auto stager = [&](int chunk, int buf) {
    float *xs = lds + buf * C::STAGE;
    return ConvStager<C>{inp + chunk * xstep, wp + chunk * wstep + cout0, xs, xs + C::XSZ, xoffs, woffs, tid, wave, tail,
                         vin.cs, chunk < nchunk, tail != 0 && chunk == nchunk - 1};
  };

  // per-lane LDS byte addresses of the operand fragments in buffer 0 (see conv_read_pair)
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);  // low half of the flat address = LDS offset
  unsigned abase = lds0 + (unsigned)(C::XSZ + half * C::BM + wm * MT * 32 + l31) * 4u;
  unsigned bbase[NT][C::ND];
#pragma unroll
  for (int n = 0; n < NT; n++) {
    const int seg = wn * NT + n;  // row = seg / 2, 32-column half = seg % 2 of the workgroup tile
    const int prow = seg >> 1, pcol = (seg & 1) * 32 + l31;
#pragma unroll
    for (int d = 0; d < C::ND; d++)
      bbase[n][d] = lds0 + (unsigned)(prow * P::Q * P::PC + pcol * P::Q + half * C::delta(d)) * 4u;
  }
  static_assert(C::STAGE * 4 <= 65536, "operand offsets of one buffer fit the 16-bit immediate");
  static_assert(kAhead * (MT + NT) <= 15, "lgkmcnt counts to 15");

  // bias / PReLU slope of cout `tid` of the block: requested now, parked in LDS after the matrix loop
  float my_bias = 0.f, my_slope = 0.f;
  if (tid < C::BM) {
    const int co = cout0 + tid < cout ? cout0 + tid : cout - 1;
    if (ep.bias) my_bias = ep.bias[co];
    if (ep.act == 1) my_slope = ep.slope[co];
  }

  stager(0, 0).issue_all();
  __syncthreads();  // (also waits for the DMA: vmcnt(0))
  CONV_STAMP(st1);

  for (int chunk = 0; chunk < nchunk; chunk++) {
    // the other buffer's last readers finished before the barrier that ended the
    // previous iteration; the DMA has the whole MFMA block to land
    stager(chunk + 1, (chunk & 1) ^ 1).issue_all();
    float a[kAhead + 1][MT], b[kAhead + 1][NT];
    conv_chunk_prologue<C, 0>(a, b, abase, bbase);
    conv_chunk_step<C, SQ, 0>(acc, a, b, abase, bbase);
    // on to the other buffer
    const unsigned flip = (chunk & 1) ? (unsigned)(-C::STAGE * 4) : (unsigned)(C::STAGE * 4);
    abase += flip;
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int d = 0; d < C::ND; d++) bbase[n][d] += flip;
    __syncthreads();
  }

  CONV_STAMP(st2);
  // (the barrier that ended the last chunk: nobody reads the stage memory any more)
  static_assert(2 * C::BM <= C::STAGE && C::BM <= kThreads, "bias / slope table fits the stage memory");
  if (tid < C::BM) {
    lds[tid] = my_bias;
    lds[C::BM + tid] = my_slope;
  }
  __syncthreads();
  {
    if constexpr (WAY == 0) {
      conv_epilogue<MT, NT, WN, (KS == 1 && S == 1 && MT == 3) ? PCONV_1X1_EPI_ROWS : 16>(acc, ep, inp, outp, vin, vout, t, r0, c0, cout0,
                                                                                          cout, ho, wo, wm, wn, l31, half, lds, lds + C::BM);
    } else if constexpr (WAY == 5) {
      // depth-to-width store in quads for full tiles
      static_assert(WAY != 5 || (MT == 3 && NT == 1 && kThreads == 512 && !SQ), "d2w quad way out: 96 couts x 32 pixels per wave");
      static_assert(WAY != 5 || (2 * C::BM + 32 * WM * 32 * WN) <= 2 * C::STAGE, "a round fits the stage memory behind the tables");
      if (c0 + kTileCols <= wo && r0 + kTileRows <= ho && cout0 + C::BM <= cout)
        conv_epilogue_quads_d2w<MT, WM, WN>(acc, ep, outp, vout, r0, c0, cout0, wm, wn, l31, half, lds + 2 * C::BM, lds, lds + C::BM, tid);
      else
        conv_epilogue<MT, NT, WN, PCONV_QUAD_FALLBACK_ROWS>(acc, ep, inp, outp, vin, vout, t, r0, c0, cout0, cout, ho, wo, wm, wn, l31, half,
                                                           lds, lds + C::BM);
    } else if constexpr (WAY >= 3) {
      // quads through the stage memory for full tiles, the pipelined element-wise way out for the ragged ones
      static_assert(WAY < 3 || WAY == 5 || (MT == 3 && NT == 1 && kThreads == 512 && (!SQ || (KS == 1 && S == 1))), "quad way out: 96 couts x 32 pixels per wave, eight waves");
      static_assert(WAY < 3 || WAY == 5 || (2 * C::BM + 32 * WM * 32 * WN) <= 2 * C::STAGE, "a round fits the stage memory behind the tables");
      if (c0 + kTileCols <= wo && cout0 + C::BM <= cout)
        conv_epilogue_quads<MT, WM, WN, SQ, WAY == 4>(acc, ep, inp, outp, vin, vout, t, r0, c0, cout0, ho, wo, wm, wn, l31,
                                                      half, lds + 2 * C::BM, lds, lds + C::BM, tid);
      else
        conv_epilogue<MT, NT, WN, PCONV_QUAD_FALLBACK_ROWS>(acc, ep, inp, outp, vin, vout, t, r0, c0, cout0, cout, ho, wo, wm, wn, l31, half,
                                                           lds, lds + C::BM);
    } else
      conv_epilogue_pipe<MT, NT, WN, SQ, WAY == 2>(acc, ep, inp, outp, vin, vout, t, r0, c0, cout0, cout, ho, wo, wm, wn,
                                                  l31, half, lds, lds + C::BM);
  }
#ifdef PCONV_CONV_STAMP
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stores have left)
    CONV_STAMP(st3);
    // sixty-four consecutive workgroups from the middle of the grid: who runs when, on which CU
    const int slot = (int)blockIdx.x - (int)(gridDim.x / 2);
    if (slot >= 0 && slot < 64 && lane == 0) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned long long *o = conv_stamps[slot][wave & 7];
      o[0] = st0, o[1] = st1, o[2] = st2, o[3] = st3, o[4] = hw, o[5] = blockIdx.x;
    }
  }
#endif
}

// ---- weight-resident, register-blocked 1x1 convolution ---------------------------------
// The 1x1 layers (ResidualBlock conv1 / conv3, the attention gate, the up-sampling
// shortcut, the GDN contraction) have K = cin <= 192.  In the tiled kernel above their
// matrix loop is short (8 k-pairs per chunk, 6-12 chunks) next to what surrounds it: a
// barrier and an LDS-DMA round per chunk and an epilogue of 48 four-byte loads / stores per
// lane -- the matrix pipes are 20-45 % busy.  This kernel is built the other way round:
//   * the WHOLE [K][BM] weight slab of the workgroup's cout block sits in LDS (<= 144 KB,
//     staged once, one workgroup per CU) and the reduction loop has no barrier;
//   * a wave owns 96 couts x 128 pixels: 3 x 4 accumulator tiles.  Lane l of either half
//     holds 4 CONSECUTIVE pixels (4 l31 .. 4 l31 + 3) of one row: its B operands of a k-pair
//     are ONE 16-byte load straight from global memory (channel 2kp for lanes 0-31, 2kp + 1
//     for lanes 32-63; 2 x 512 contiguous bytes per wave instruction) through a register
//     ring 6 deep -- 6 KB in flight per wave, 48 KB per CU -- and its results leave as
//     16-byte stores (gate / residual / GDN input arrive as 16-byte loads): 12 instead of 48
//     memory instructions per lane on the way out;
//   * 12 MFMAs per k-pair against 3 LDS reads and 1 global load.
// A workgroup = 8 waves = WM cout halves x (8 / WM) pixel groups, tile = ROWS rows x 256
// columns, `passes` such tiles with the same slab.  Lanes past the right edge take the last
// 4 columns instead (w - 4 .. w - 1): they recompute and rewrite their neighbours' values,
// bit for bit, so there is no ragged path.  Per output the MFMA sequence is the tiled
// kernel's (k ascending from 0) and so is the epilogue arithmetic: identical bits.
typedef const __attribute__((address_space(1))) char global_bytes;  // (a global, not a flat, access)
__device__ __forceinline__ float4 load4_base_off(global_bytes *base, unsigned off) {
  asm volatile("" : "+v"(off));  // keeps the 32-bit lane offset out of a hoisted 64-bit add
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = *reinterpret_cast<const __attribute__((address_space(1))) v4f *>(base + off);
  return make_float4(v.x, v.y, v.z, v.w);
}

// RES: the layer adds a residual.  The kernel takes the epilogues of the codec's big 1x1 layers: bias, PReLU
// (`act` 0 = PReLU with slope 1: v * 1 is v), the GDN pair (SQ), residual, trim; gates, the sigmoid and the
// depth-to-width store stay with the tiled kernel.
template <int WM, bool SQ, bool RES>
__global__ __launch_bounds__(512, 2) void conv1x1_rb_kernel(
    const float *__restrict__ in, const float *__restrict__ wp, float *__restrict__ out, int kpad, int h, int w,
    int cout, int cout_pad, int tiles_r, int tiles_c, int cblocks, int passes, int stagger, ConvView vin,
    ConvView vout, ConvEpilogue ep) {
  constexpr int kThreads = 512, MT = 3, NT = 4, BM = 96 * WM, WP = 8 / WM, ROWS = WP / 2, D = 5;
  constexpr int kCols = 256;
  extern __shared__ float lds[];  // ws[kpad][BM]
  int b = blockIdx.x;
  const int cb = b % cblocks;
  b /= cblocks;
  const int trx = b % tiles_r;
  b /= tiles_r;
  const int tcx = b % tiles_c;
  const int t = b / tiles_c;
  const int cout0 = cb * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wq = wave / WM;  // cout half, pixel group
  const int l31 = lane & 31, half = lane >> 5;
  float *outp = out + (size_t)t * vout.ts;
  const float *inp = in + (size_t)t * vin.ts;
  const int gc0 = tcx * kCols + (wq & 1) * 128;  // first column of the wave's pixel group
  const int limit = ep.col_limit ? ep.col_limit[t % ep.npart] : w;
  // the tiled kernel writes zeros from the first 64-column block that starts in the dead
  // columns on; the same columns are zero here, whatever the epilogue
  const int dead_at = ep.col_limit ? (limit + 63) / 64 * 64 : w;
  const bool tile_dead = tcx * kCols >= dead_at;
  if (!tile_dead) {
    typedef __attribute__((address_space(3))) void lds_ptr_t;
    typedef const __attribute__((address_space(1))) void glb_ptr_t;
    const int n4 = kpad * BM / 4;
    for (int e4 = tid; e4 < n4; e4 += kThreads) {
      const int kk = e4 / (BM / 4), co = (e4 % (BM / 4)) * 4;
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)(wp + (size_t)kk * cout_pad + cout0 + co),
                                       (lds_ptr_t *)(lds + (size_t)(e4 - lane) * 4), 16, 0, 0);
    }
    // bias and PReLU slope of the cout block, behind the slab (see conv_epilogue: loaded row by row in
    // front of their use they sat behind the previous row's store)
    if (tid < BM) {
      const int co = cout0 + tid < cout ? cout0 + tid : cout - 1;
      lds[kpad * BM + tid] = ep.bias ? ep.bias[co] : 0.f;
      lds[kpad * BM + BM + tid] = ep.act == 1 ? ep.slope[co] : 1.f;
    }
    __syncthreads();  // (waits for the DMA: vmcnt(0)); the only barrier of the kernel
  }
  const float *bias_s = lds + kpad * BM, *slope_s = bias_s + BM;
  if (gc0 >= w) return;  // the group lies past the right edge of the tensor
  const int KP = kpad / 2;
  const size_t kstep = (size_t)2 * vin.cs * sizeof(float);
  const float *wsl = lds + half * BM + wm * 96 + l31;
  // this lane's 4 columns (the last 4 of the row for lanes past the edge)
  const int col = gc0 + 4 * l31 < w - 4 ? gc0 + 4 * l31 : w - 4;
  const bool group_dead = gc0 >= dead_at;
  // Waves w and w + 4 share a SIMD.  Started together they stay together: both in the matrix loop
  // (taking turns on the one matrix pipe), then both in the way out (pipe idle, 3 x 16-byte accesses
  // per 4 outputs each).  Holding the second wave back by about one matrix loop puts one of them on
  // the pipe while the other moves data.  (The pair that shares its B operands, waves 2q and 2q + 1,
  // sits in the same half and keeps its phase.)
  if (wave >= 4)
    for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(127);
#pragma unroll 1
  for (int pass = 0; pass < passes; pass++) {
    const int orow = (trx * passes + pass) * ROWS + (wq >> 1);
    if (orow >= h) break;
    // (cout0 made opaque per pass: otherwise the per-channel bias / slope loads of the
    // epilogue are hoisted out of the pass loop and live across the matrix loop)
    int cbase = __builtin_amdgcn_readfirstlane(cout0 + wm * 96);
    asm volatile("" : "+s"(cbase));
    if (group_dead) {
      // whole group in the dead columns of this latitude band: zeros (128 columns of a cout
      // row = 64 lanes x 2 columns)
      const int c2 = gc0 + 2 * lane;
#pragma unroll 1
      for (int e = 0; e < 96; e++) {
        const int co = cbase + e;
        if (co >= cout) break;
        for (int j = 0; j < 2; j++) {
          if (c2 + j >= w) continue;
          outp[(size_t)co * vout.cs + (size_t)orow * vout.rs + c2 + j] = 0.f;
        }
      }
      continue;
    }
    const unsigned loff = (unsigned)(((long long)half * vin.cs + (long long)orow * vin.rs + col) * sizeof(float));
    // wave-uniform base in scalar registers: the loads are "SGPR base + lane offset"
    const unsigned long long in_u = reinterpret_cast<unsigned long long>(inp);
    global_bytes *bbase = reinterpret_cast<global_bytes *>(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(in_u >> 32)) << 32) |
        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)in_u));  // (int result: no sign extension)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int n = 0; n < NT; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    float4 bq[D];
    float a[2][MT];
#pragma unroll
    for (int j = 0; j < D; j++) bq[j] = load4_base_off(bbase + (size_t)(j < KP ? j : KP - 1) * kstep, loff);
#pragma unroll
    for (int m = 0; m < MT; m++) a[0][m] = wsl[m * 32];
    // one round = U k-pairs: a multiple of the ring depth (static ring slots) and of 2 (the
    // two A fragment sets alternate)
    constexpr int U = (D % 2) ? 2 * D : D;
#pragma unroll 1
    for (int kp0 = 0; kp0 < KP; kp0 += U) {
#pragma unroll
      for (int j = 0; j < U; j++) {
        const int kp = kp0 + j;
        if (kp < KP) {  // (the last round may be partial)
          float4 bv = bq[j % D];
          // the ring always refills (past the end: the last k-pair again, never used) and the
          // next A fragment is always read: no data-dependent branch inside the matrix loop
          const int kpre = kp + D < KP ? kp + D : KP - 1;
          bq[j % D] = load4_base_off(bbase + (size_t)kpre * kstep, loff);
          const int knext = kp + 1 < KP ? kp + 1 : KP - 1;
#pragma unroll
          for (int m = 0; m < MT; m++) a[(j + 1) & 1][m] = wsl[(size_t)knext * 2 * BM + m * 32];
          if (SQ) {
            bv.x *= bv.x;
            bv.y *= bv.y;
            bv.z *= bv.z;
            bv.w *= bv.w;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MT; m++) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][m], bv.x, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][m], bv.y, acc[m][1], 0, 0, 0);
            acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][m], bv.z, acc[m][2], 0, 0, 0);
            acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][m], bv.w, acc[m][3], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- way out: 4 consecutive pixels of one cout row per step, 48 rows per lane ----
    // What a row reads from memory (the residual, the GDN's own input) is requested a batch of rows
    // ahead, with addresses clamped into the tensor instead of guards around the loads: straight-line
    // code, so the compiler counts the loads and the stores (vmcnt(n)) and a batch's wait leaves the next
    // batch and the stores of the one before in flight.  (Loaded row by row in front of their use, behind
    // a guard, every row was load -> vmcnt(0) -> store: 48 serial round trips per pass.)
    const int act = ep.act;
    const int trim_at = ((ep.trim || act == 2 || act == 3) && ep.col_limit) ? limit : dead_at;
    const float *resp = RES ? ep.residual + (size_t)t * ep.vres.ts : nullptr;
    constexpr int NROW = MT * 16, EB = (SQ && RES) ? 1 : 4, NB = NROW / EB;
    struct RowIn {
      float4 r, x;
    };
    auto row_cout = [&](int q) { return cbase + (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * half; };
    auto request = [&](int q) {
      RowIn v = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      int co = row_cout(q);
      co = co < cout ? co : cout - 1;
      if (RES) v.r = *reinterpret_cast<const float4 *>(resp + (size_t)co * ep.vres.cs + (size_t)orow * ep.vres.rs + col);
      if (SQ) v.x = *reinterpret_cast<const float4 *>(inp + (size_t)co * vin.cs + (size_t)orow * vin.rs + col);
      return v;
    };
    RowIn cur[EB], nxt[EB];
    if (RES || SQ) {
#pragma unroll
      for (int e = 0; e < EB; e++) cur[e] = request(e);
    }
#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
      if ((RES || SQ) && bi + 1 < NB) {
#pragma unroll
        for (int e = 0; e < EB; e++) nxt[e] = request((bi + 1) * EB + e);
      }
#pragma unroll
      for (int e = 0; e < EB; e++) {
        const int q = bi * EB + e, m = q >> 4, r = q & 15;
        const int co = row_cout(q);
        if (co < cout) {
          const float bco = bias_s[co - cout0], sl = slope_s[co - cout0];
          float v[4] = {acc[m][0][r] + bco, acc[m][1][r] + bco, acc[m][2][r] + bco, acc[m][3][r] + bco};
          if (SQ) {
            // 1x1, stride 1: input and output share their geometry
            const float xs[4] = {cur[e].x.x, cur[e].x.y, cur[e].x.z, cur[e].x.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float nrm = sqrtf(v[j]);
              v[j] = act == 2 ? xs[j] / nrm : xs[j] * nrm;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (v[j] < 0) v[j] = v[j] * sl;
          }
          if (RES) {
            v[0] = cur[e].r.x + v[0];
            v[1] = cur[e].r.y + v[1];
            v[2] = cur[e].r.z + v[2];
            v[3] = cur[e].r.w + v[3];
          }
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (col + j >= trim_at) v[j] = 0.f;
          *reinterpret_cast<float4 *>(outp + (size_t)co * vout.cs + (size_t)orow * vout.rs + col) =
              make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      if ((RES || SQ) && bi + 1 < NB) {
#pragma unroll
        for (int e = 0; e < EB; e++) cur[e] = nxt[e];
      }
    }
  }
}

template <int WM, bool SQ, bool RES>
int launch_conv1x1(const float *in, const float *wp, float *out, int tn, int cin, int h, int w, int cout,
                   int cout_pad, const ConvView &vin, const ConvView &vout, const ConvEpilogue &ep,
                   hipStream_t stream) {
  constexpr int BM = 96 * WM, ROWS = (8 / WM) / 2;
  const int kpad = (cin + 15) / 16 * 16;
  const int tiles_c = (w + 255) / 256;
  const int cblocks = (cout + BM - 1) / BM;
  // several row groups per workgroup (the slab is staged once for all of them) while that
  // still leaves >= 4 workgroups per CU
  const long long single = (long long)tn * ((h + ROWS - 1) / ROWS) * tiles_c * cblocks;
  const int passes = single >= 4096 ? 4 : (single >= 2048 ? 2 : 1);
  const int tiles_r = (h + ROWS * passes - 1) / (ROWS * passes);
  const long long grid = (long long)tn * tiles_r * tiles_c * cblocks;
  if (grid <= 0 || grid > 0x7fffffffLL) {
    pconv_set_error("conv2d: grid %lld out of range", grid);
    return PCONV_EINVAL;
  }
  const size_t smem = ((size_t)kpad * BM + 2 * BM) * sizeof(float);  // slab + bias / slope table
  auto kern = conv1x1_rb_kernel<WM, SQ, RES>;
  if (smem > 64 * 1024) {
    static std::atomic<unsigned long long> raised{0};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    const unsigned long long bit = 1ULL << (device & 63);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) {
        pconv_set_error("conv2d: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e));
        return PCONV_ELAUNCH;
      }
      raised.fetch_or(bit, std::memory_order_release);
    }
  }
  // second wave of every SIMD held back (units of 127 x 64 cycles); PCONV_CONV1X1_STAGGER: experiment knob
  static const int stagger_env = getenv("PCONV_CONV1X1_STAGGER") ? atoi(getenv("PCONV_CONV1X1_STAGGER")) : -1;
  const int stagger = stagger_env >= 0 ? stagger_env : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, in, wp, out, kpad, h, w, cout, cout_pad,
                     tiles_r, tiles_c, cblocks, passes, stagger, vin, vout, ep);
  return PCONV_OK;
}

// the weight-resident form takes 1x1 stride-1 layers whose slab fits LDS and that are big
// enough to fill the chip with 512-pixel workgroup tiles; 16-byte accesses need the views'
// strides and bases to be multiples of 2 floats... they only need 4-byte alignment on
// gfx950 (unaligned dwordx4 access is enabled), so any view qualifies
// (PCONV_CONV1X1=tiled forces the tiled kernel: A/B measurements and the parity tests)
inline bool use_resident_1x1(int cin, int cout, int tn, int h, int w) {
  const char *env = getenv("PCONV_CONV1X1");
  if (env && env[0] == 't') return false;
  const int kpad = (cin + 15) / 16 * 16;
  const int bm = cout > 96 ? 192 : 96;
  if (!(cin >= 32 && cin % 16 == 0 && cout > 32 && ((size_t)kpad * bm + 2 * bm) * sizeof(float) <= 160 * 1024)) return false;
  if (w < 4) return false;
  if (env && env[0] == 'r') return true;  // forced (tests, A/B measurements)
  // Measured (MI355X, layers of a 4096x2048 frame).  Against the tiled kernel of round 2's first
  // half this form gained 10-25 % on 96->192, 192->192 and the GDN.  Since the tiled kernel reads
  // its LDS operands with counted waits and requests what its way out reads in batches, it is the
  // faster one everywhere: 96->192 + residual 0.36 vs 0.41 ms, 192->192 0.50 vs 0.56, 192->768
  // 1.51 vs 1.59, GDN at 2048 columns 2.14 vs 2.63 ms.  The resident form stays as the measured
  // alternative (PCONV_CONV1X1=resident).
  (void)tn, (void)h;
  return false;
}

// ---- 3x3 stride-1 convolution with <= 16 couts on v_mfma_f32_16x16x4_f32 --------------------
// The codec's last layer (model_zoo_v2.py:296-303: 192 -> 12 channels, then Dtow to the 3 image planes) is the one
// place where the 32-cout granule of v_mfma_f32_32x32x2_f32 hurts: 12 of 32 matrix rows do work (the 32-cout tile
// of conv_mfma_kernel runs it at 0.28 of the peak in useful flops, 0.375 is its ceiling).  The 16 x 16 x 4
// instruction has half the rows at the same rate -- 12 of 16 busy -- and accumulates its four products as one
// k-ascending fmaf chain just like the 32 x 32 x 2 one (tools/mfma16_chain_probe.hip: 0 of 256 outputs differ), so
// the numerics contract (one k-ascending chain per output) holds bit for bit.
//   workgroup = 8 waves = 8 output rows x 64 columns, all 16 couts; wave = one row = 4 column groups of 16;
//   per chunk of 4 input channels (36 reduction entries = 9 instructions of 4): patch 4 x 10 x 66 and weights
//   36 x 16 by LDS-DMA, double buffered; lane (r = l % 16, q = l / 16) feeds entry 4 s + q: A = W[4 s + q][r],
//   B = patch element of entry 4 s + q at the wave's row, column 16 n + r; D: couts 4 q .. 4 q + 3 of column r.
// With the depth-to-width store a lane holds exactly the 2 x 2 sub-pixels of output channel q: two 8-byte stores.
// Takes bias, PReLU, trim / dead tiles, the Dtow store; no residual / gate / sigmoid.
template <int KS, int KC>
__global__ __launch_bounds__(512, 2) void conv_small_kernel(const float *__restrict__ in, const float *__restrict__ wp,
                                                            float *__restrict__ out, int cin, int h, int w, int cout,
                                                            int cout_pad, int ho, int wo, int tiles_r, int tiles_c,
                                                            ConvView vin, ConvView vout, ConvEpilogue ep) {
  static_assert(KS == 3, "small-cout kernel: 3x3 stride 1");
  constexpr int ROWS = 8, PR = ROWS + KS - 1, PC = kTileCols + KS - 1, KK = KC * KS * KS, NS = KK / 4;
  constexpr int XSZ = KC * PR * PC, WSZ = KK * 16, STAGE = XSZ + WSZ, kThreads = 512;
  constexpr int XLD = (XSZ + kThreads - 1) / kThreads;
  static_assert(KK % 4 == 0 && KK * 4 <= kThreads && (2 * STAGE) * 4 + 256 < 65536, "immediate offsets reach both buffers");
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
  int b = blockIdx.x;
  const int trx = b % tiles_r;
  b /= tiles_r;
  const int tcx = b % tiles_c;
  const int t = b / tiles_c;
  const int r0 = trx * ROWS, c0 = tcx * kTileCols;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  float *outp = out + (size_t)t * vout.ts;
  if (ep.col_limit && c0 >= ep.col_limit[t % ep.npart]) {
    conv_zero_tile<16, ROWS, kThreads>(outp, vout, ep.d2w, 0, cout, r0, c0, ho, wo, tid);
    return;
  }
  const float *inp = in + (size_t)t * vin.ts;
  const int nchunk = (cin + KC - 1) / KC, tail = cin % KC;
  unsigned xoffs[XLD];
#pragma unroll
  for (int j = 0; j < XLD; j++) {
    int e = tid + j * kThreads;
    e = e < XSZ ? e : 0;
    const int pc = e % PC, pr = (e / PC) % PR, ci = e / (PC * PR);
    int ir = r0 + pr, ic = c0 + pc;
    ir = ir < h ? ir : h - 1;
    ic = ic < w ? ic : w - 1;
    xoffs[j] = (unsigned)((ci * vin.cs + (long long)ir * vin.rs + ic) * 4);
  }
  // weights: rows of the packed [k][cout_pad] slab, the first 16 couts: 4 float4 per reduction entry
  const int we4 = tid < KK * 4 ? tid : 0;
  const unsigned woff = (unsigned)(((we4 >> 2) * cout_pad + (we4 & 3) * 4) * 4);
  const size_t xstep = (size_t)KC * vin.cs, wstep = (size_t)KK * cout_pad;
  auto stage = [&](int chunk, int buf) {
    if (chunk >= nchunk) return;
    float *xs = lds + buf * STAGE;
    const float *xb = inp + chunk * xstep;
    const bool ragged = tail != 0 && chunk == nchunk - 1;
#pragma unroll
    for (int j = 0; j < XLD; j++) {
      const int e = tid + j * kThreads;
      if (e < XSZ) {
        unsigned off = xoffs[j];
        if (ragged) {
          // channels past cin: the last real one again -- their weight rows are zero: + 0 to the chain
          const int ci = e / (PC * PR);
          if (ci >= tail) off -= (unsigned)((ci - tail + 1) * vin.cs * 4);
        }
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_global_load_lds((glb_ptr_t *)((glb_bytes_t *)xb + off), (lds_ptr_t *)(xs + j * kThreads + wave * 64), 4, 0,
                                         0);
      }
    }
    if (tid < KK * 4) {
      unsigned off = woff;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)((glb_bytes_t *)(wp + chunk * wstep) + off),
                                       (lds_ptr_t *)(xs + XSZ + wave * 64 * 4), 16, 0, 0);
    }
  };
  // the lane's patch element of entry 4 s + q, at the wave's row, column r (column group n: + 16 n floats)
  int boff[NS];
#pragma unroll
  for (int s4 = 0; s4 < NS; s4++) {
    const int k = 4 * s4 + q;
    boff[s4] = (k / (KS * KS)) * PR * PC + ((k / KS) % KS) * PC + k % KS + wave * PC + r;
  }
  const int aoff = XSZ + q * 16 + r;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // operand reads by hand, as in conv_mfma_kernel: `ds_read_b32 dst, base offset:imm`, the five operands of
  // instruction group s + 1 requested before the four MFMAs of group s, counted waits (the compiler's own
  // schedule waited for every pair of reads in front of the two MFMAs that use it)
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);
  const unsigned abase = lds0 + (unsigned)aoff * 4u;
  unsigned bbase[NS];
#pragma unroll
  for (int s4 = 0; s4 < NS; s4++) bbase[s4] = lds0 + (unsigned)boff[s4] * 4u;
  auto chunk_body = [&](auto bufc) {
    constexpr int BO = decltype(bufc)::value * STAGE * 4;
    float a[2], bq[2][4];
    a[0] = lds_read_imm<BO>(abase);
    static_for_<0, 4>([&](auto nc) { bq[0][decltype(nc)::value] = lds_read_imm<BO + 64 * decltype(nc)::value>(bbase[0]); });
    static_for_<0, NS>([&](auto sc) {
      constexpr int S4 = decltype(sc)::value;
      if constexpr (S4 + 1 < NS) {
        a[(S4 + 1) & 1] = lds_read_imm<BO + (S4 + 1) * 256>(abase);
        static_for_<0, 4>([&](auto nc) {
          bq[(S4 + 1) & 1][decltype(nc)::value] = lds_read_imm<BO + 64 * decltype(nc)::value>(bbase[S4 + 1]);
        });
      }
      float &A = a[S4 & 1];
      float(&B)[4] = bq[S4 & 1];
      asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(A), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]) : "n"(S4 + 1 < NS ? 5 : 0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 4; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B[n], acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  stage(0, 0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    stage(chunk + 1, 1);
    chunk_body(std::integral_constant<int, 0>{});
    __syncthreads();
    if (chunk + 1 >= nchunk) break;
    stage(chunk + 2, 0);
    chunk_body(std::integral_constant<int, 1>{});
    __syncthreads();
  }
  // ---- way out: lane (r, q) holds couts 4 q .. 4 q + 3 of row `wave`, columns 16 n + r ----
  const int orow = r0 + wave;
  if (orow >= ho) return;
  const int act = ep.act;
  const int trim_at = (ep.trim && ep.col_limit) ? ep.col_limit[t % ep.npart] : wo;
  float bco[4], sl[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int co = 4 * q + i < cout ? 4 * q + i : cout - 1;
    bco[i] = ep.bias ? ep.bias[co] : 0.f;
    sl[i] = act == 1 ? ep.slope[co] : 1.f;
  }
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int ocol = c0 + 16 * n + r;
    if (ocol >= wo) continue;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[i] = acc[n][i] + bco[i];
      if (act == 1 && v[i] < 0) v[i] = v[i] * sl[i];
    }
    if (ep.d2w) {
      // cout 4 q + 2 sy + sx -> channel q, row 2 orow + sy, column 2 ocol + sx
      if (4 * q + 3 < cout) {
#pragma unroll
        for (int sy = 0; sy < 2; sy++)
          *reinterpret_cast<float2 *>(outp + (size_t)q * vout.cs + (size_t)(2 * orow + sy) * vout.rs + 2 * ocol) =
              make_float2(v[2 * sy], v[2 * sy + 1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (4 * q + i < cout) outp[(size_t)(4 * q + i) * vout.cs + (size_t)orow * vout.rs + ocol] = ocol >= trim_at ? 0.f : v[i];
    }
  }
}

// PCONV_CONV_SMALL=0: the 32-cout tile of conv_mfma_kernel keeps the layers with <= 16 couts (A/B, parity test)
inline bool use_small_cout() {
  const char *env = getenv("PCONV_CONV_SMALL");  // (read per call: the parity test switches it)
  return !(env && env[0] == '0');
}

int launch_conv_small(const float *in, const float *wp, float *out, int tn, int cin, int h, int w, int cout, int cout_pad,
                      int ho, int wo, const ConvView &vin, const ConvView &vout, const ConvEpilogue &ep, hipStream_t stream) {
  const int tiles_r = (ho + 7) / 8, tiles_c = (wo + kTileCols - 1) / kTileCols;
  const long long grid = (long long)tn * tiles_r * tiles_c;
  if (grid <= 0 || grid > 0x7fffffffLL) {
    pconv_set_error("conv2d: grid %lld out of range", grid);
    return PCONV_EINVAL;
  }
  // (four input channels per stage: eight -- half the chunk barriers, two workgroups per CU instead of three --
  // measured level, 1.013-1.021 vs 1.005-1.014 ms)
  hipLaunchKernelGGL((conv_small_kernel<3, 4>), dim3((unsigned)grid), dim3(512), 0, stream, in, wp, out, cin, h, w, cout,
                     cout_pad, ho, wo, tiles_r, tiles_c, vin, vout, ep);
  return PCONV_OK;
}

#ifdef PCONV_CONV_STAMP
extern "C" int pconv_conv_read_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(conv_stamps), sizeof(conv_stamps)) == hipSuccess ? 0 : 1;
}
#endif

// PCONV_CONV1X1_WAYOUT=batch: the 1x1 / GDN layers keep conv_epilogue (A/B measurements)
inline bool pipe_way_out() {
  const char *env = getenv("PCONV_CONV1X1_WAYOUT");  // (read per call: the parity test switches it)
  return !(env && env[0] == 'b');
}
// default: full tiles leave through the stage memory in 16-byte quads (conv_epilogue_quads);
// PCONV_CONV1X1_WAYOUT=pipe / batch: the element-wise ways out (A/B measurements, parity tests)
inline bool quad_way_out() {
  const char *env = getenv("PCONV_CONV1X1_WAYOUT");
  return !(env && (env[0] == 'p' || env[0] == 'b'));
}

// (cout, cin, k, k) -> [k_pad][cout_pad], k = (ci*KS + kh)*KS + kw, zero padded
__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ packed, int cout,
                                   int red, int cout_pad, int red_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout_pad * red_pad) return;
  const int co = i % cout_pad, kk = i / cout_pad;
  packed[i] = (co < cout && kk < red) ? w[(size_t)co * red + kk] : 0.f;
}

template <int MT, int NT, int WM, int WN, int KS, int S, int KC, bool SQ = false, int WAY = 0>
int launch_conv(const float *in, const float *wp, float *out, int tn, int cin, int h, int w, int cout,
                int cout_pad, int ho, int wo, const ConvView &vin, const ConvView &vout, const ConvEpilogue &ep,
                hipStream_t stream) {
  using C = ConvCfg<MT, NT, WM, WN, KS, S, KC>;
  constexpr int kThreads = C::THREADS;
  constexpr int kTileRows = C::ROWS;
  const int tiles_r = (ho + kTileRows - 1) / kTileRows;
  const int tiles_c = (wo + kTileCols - 1) / kTileCols;
  const int cblocks = (cout + C::BM - 1) / C::BM;
  const long long grid = (long long)tn * tiles_r * tiles_c * cblocks;
  if (grid <= 0 || grid > 0x7fffffffLL) {
    pconv_set_error("conv2d: grid %lld out of range", grid);
    return PCONV_EINVAL;
  }
  const size_t smem = (size_t)2 * C::STAGE * sizeof(float);
  auto kern = conv_mfma_kernel<MT, NT, WM, WN, KS, S, KC, SQ, WAY>;
  if (smem > 64 * 1024) {
    // the dynamic-LDS limit is a per-device attribute of the function: raise it once on
    // every device this process launches the instantiation on (one process may drive
    // several GPUs: nn.DataParallel replicas of BaseOpModule)
    static std::atomic<unsigned long long> raised{0};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    const unsigned long long bit = 1ULL << (device & 63);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) {
        pconv_set_error("conv2d: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e));
        return PCONV_ELAUNCH;
      }
      raised.fetch_or(bit, std::memory_order_release);
    }
  }
  // PCONV_CONV_XCD=1 turns the XCD-grouped order on.  Measured on the analysis transform of a
  // 4096x2048 frame (same box, alternating runs): HBM reads of the 3x3 192-cout kernel 1.46 GB per
  // launch instead of 1.82 GB (FETCH_SIZE, profiles/), but 59.3-59.5 ms per frame instead of
  // 58.4-58.5: the kernel is not bandwidth-bound (0.3 TB/s) and the grouped order starts the
  // stripes of a tile one after the other on an XCD.  Off by default; the order is kept as the
  // measured alternative.
  static const bool xcd_order = getenv("PCONV_CONV_XCD") && atoi(getenv("PCONV_CONV_XCD")) == 1;
  const int xcd_group = (xcd_order && KS == 3) ? cblocks * tiles_r : 0;  // 1x1 tiles share no rows
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kThreads), smem, stream, in, wp, out, cin, h, w, cout,
                     cout_pad, ho, wo, tiles_r, tiles_c, cblocks, xcd_group, vin, vout, ep);
  return PCONV_OK;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline ConvView dense_view(int c, int h, int w) { return {(long long)c * h * w, (long long)h * w, w}; }

// views[3*i .. 3*i+2] = (tile, channel, row) strides of tensor i, or a dense view when views is null
inline ConvView view_at(const long long *views, int i, int c, int h, int w) {
  if (!views) return dense_view(c, h, w);
  return {views[3 * i], views[3 * i + 1], (int)views[3 * i + 2]};
}

inline bool view_ok(const ConvView &v, int c, int h, int w) {
  return v.rs >= w && v.cs >= (long long)(h - 1) * v.rs + w && v.ts >= (long long)(c - 1) * v.cs + (long long)(h - 1) * v.rs + w;
}

}  // namespace

extern "C" int pconv_conv_packed_size(int cout, int cin, int k, int *cout_pad, int *red_pad) {
  // cout padded to the widest workgroup tile, reduction to whole chunks
  const int cp = round_up(cout, cout > 96 ? 192 : (cout > 32 ? 96 : 32));
  const int kc = (k == 1) ? PCONV_KC1 : 4;
  const int rp = round_up(cin, kc) * k * k;
  if (cout_pad) *cout_pad = cp;
  if (red_pad) *red_pad = rp;
  return cp * rp;
}

extern "C" int pconv_conv_pack_weight(const float *w, float *packed, int cout, int cin, int k,
                                      void *stream) {
  PCONV_REQUIRE(w && packed && cout > 0 && cin > 0 && (k == 1 || k == 3), "conv_pack: bad argument");
  int cp, rp;
  const int total = pconv_conv_packed_size(cout, cin, k, &cp, &rp);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream),
                     w, packed, cout, cin * k * k, cp, rp);
  PCONV_LAUNCH_CHECK("conv_pack_weight");
  return PCONV_OK;
}

extern "C" int pconv_conv2d(const float *in, const float *packed_w, const float *bias, float *out,
                            int tn, int cin, int h, int w, int cout, int k, int stride, int act,
                            const float *slope, const int32_t *col_limit, int npart,
                            const float *residual, const float *gate, int trim, int d2w,
                            const long long *views, void *stream) {
  PCONV_REQUIRE(in && packed_w && out, "conv2d: null pointer");
  PCONV_REQUIRE(!d2w || (cout % 4 == 0 && act <= 1 && !residual && !gate && !trim),
                "conv2d: depth-to-width needs cout %% 4 == 0 and takes no sigmoid / gate / residual / trim");
  PCONV_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), "conv2d: k=%d stride=%d unsupported",
                k, stride);
  PCONV_REQUIRE(h >= k && w >= k && tn > 0 && cin > 0 && cout > 0, "conv2d: bad shape");
  PCONV_REQUIRE(act == 0 || (act == 1 && slope) || act == 4, "conv2d: bad activation %d", act);
  PCONV_REQUIRE(!col_limit || npart > 0, "conv2d: col_limit needs npart");
  PCONV_REQUIRE(!trim || col_limit, "conv2d: trim needs col_limit");
  PCONV_REQUIRE(residual != out && gate != out, "conv2d: residual / gate must not alias the output");
  const int ho = (h - k) / stride + 1, wo = (w - k) / stride + 1;
  int cp, rp;
  pconv_conv_packed_size(cout, cin, k, &cp, &rp);
  hipStream_t s = as_stream(stream);
  const int oc = d2w ? cout / 4 : cout, oh = d2w ? 2 * ho : ho, ow = d2w ? 2 * wo : wo;  // stored geometry
  const ConvView vin = view_at(views, 0, cin, h, w), vout = view_at(views, 1, oc, oh, ow);
  const ConvEpilogue ep = {bias,  slope, residual, gate, col_limit, npart, act, trim,
                           view_at(views, 2, cout, ho, wo), view_at(views, 3, cout, ho, wo), d2w};
  PCONV_REQUIRE(!d2w || (vout.rs % 2 == 0 && vout.cs % 2 == 0 && vout.ts % 2 == 0 &&
                         (reinterpret_cast<uintptr_t>(out) & 7) == 0),
                "conv2d: depth-to-width output must be 8-byte aligned row by row");
  PCONV_REQUIRE(view_ok(vin, cin, h, w) && view_ok(vout, oc, oh, ow) &&
                    (!residual || view_ok(ep.vres, cout, ho, wo)) && (!gate || view_ok(ep.vgate, cout, ho, wo)),
                "conv2d: strides overlap");
  PCONV_REQUIRE(((long long)(16 - 1) * vin.cs + (long long)(h - 1) * vin.rs + w) * 4 < (1LL << 32),
                "conv2d: input channel stride too large for 32-bit byte offsets inside a chunk");
  // the quad ways out (conv_epilogue_quads) address output and residual as a 64-bit uniform base + a 32-bit byte
  // offset per lane that spans up to 32 cout rows of a parked round: their strides are independent of the input's
  PCONV_REQUIRE(((long long)(32 - 1) * vout.cs + (long long)(oh - 1) * vout.rs + ow) * 4 < (1LL << 32) &&
                    (!residual || ((long long)(32 - 1) * ep.vres.cs + (long long)(ho - 1) * ep.vres.rs + wo) * 4 < (1LL << 32)),
                "conv2d: output / residual channel stride too large for 32-bit byte offsets inside a round");
  int rc;
#define ARGS in, packed_w, out, tn, cin, h, w, cout, cp, ho, wo, vin, vout, ep, s
  // workgroup tiles (measured on MI355X, 192->192 3x3 at 16 x 64 x 2048: 127 TFLOP/s):
  //   cout > 96 : 192 couts x (2 rows x 64 px), 8 waves of 96 x 32: 135.8 TFLOP/s (4 waves of 96 x 64 px
  //               = 3 x 2 accumulator tiles each: 128.3; 6 such waves on 3 rows x 64 px: 97)
  //   cout > 32 :  96 couts x (4 rows x 64 px), 8 waves of 96 x 32
  //   else      :  32 couts x (2 rows x 64 px), 4 waves of 32 x 32
#define BY_TILE(KS, S, KC)                                         \
  if (cout > 96)                                                   \
    rc = launch_conv<3, 1, 2, 4, KS, S, KC>(ARGS);                 \
  else if (cout > 32)                                              \
    rc = launch_conv<3, 1, 1, 8, KS, S, KC>(ARGS);                 \
  else                                                             \
    rc = launch_conv<1, 1, 1, 4, KS, S, KC>(ARGS);
  if (k == 3 && stride == 1 && cout <= 16 && !residual && !gate && act != 4 && (!d2w || cout % 4 == 0) && use_small_cout()) {
    rc = launch_conv_small(in, packed_w, out, tn, cin, h, w, cout, cp, ho, wo, vin, vout, ep, s);
  } else if (k == 3 && stride == 1) {
    BY_TILE(3, 1, 4)
  } else if (k == 3 && stride == 2 && cout > 32 && !gate && !d2w && act != 4 && quad_way_out()) {
    // (the stride-2 3x3 layers of the Down blocks: the same workgroup tiles, the same quad way out)
    if (cout > 96)
      rc = residual ? launch_conv<3, 1, 2, 4, 3, 2, 4, false, 4>(ARGS) : launch_conv<3, 1, 2, 4, 3, 2, 4, false, 3>(ARGS);
    else
      rc = residual ? launch_conv<3, 1, 1, 8, 3, 2, 4, false, 4>(ARGS) : launch_conv<3, 1, 1, 8, 3, 2, 4, false, 3>(ARGS);
  } else if (k == 3 && stride == 2) {
    BY_TILE(3, 2, 4)
  } else if (k == 1 && stride == 1 && !gate && !d2w && act != 4 && use_resident_1x1(cin, cout, tn, h, w)) {
    if (cout > 96 && residual)
      rc = launch_conv1x1<2, false, true>(in, packed_w, out, tn, cin, h, w, cout, cp, vin, vout, ep, s);
    else if (cout > 96)
      rc = launch_conv1x1<2, false, false>(in, packed_w, out, tn, cin, h, w, cout, cp, vin, vout, ep, s);
    else if (residual)
      rc = launch_conv1x1<1, false, true>(in, packed_w, out, tn, cin, h, w, cout, cp, vin, vout, ep, s);
    else
      rc = launch_conv1x1<1, false, false>(in, packed_w, out, tn, cin, h, w, cout, cp, vin, vout, ep, s);
  } else if (k == 1 && stride == 1 && d2w && cout > 32 && quad_way_out()) {
    // (d2w implies no gate / residual / sigmoid / trim: checked above)
    if (cout > 96)
      rc = launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, false, 5>(ARGS);
    else
      rc = launch_conv<3, 1, 1, 8, 1, 1, PCONV_KC1, false, 5>(ARGS);
  } else if (k == 1 && stride == 1 && !gate && !d2w && act != 4 && cout > 32 && quad_way_out()) {
    if (cout > 96)
      rc = residual ? launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, false, 4>(ARGS) : launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, false, 3>(ARGS);
    else
      rc = residual ? launch_conv<3, 1, 1, 8, 1, 1, PCONV_KC1, false, 4>(ARGS) : launch_conv<3, 1, 1, 8, 1, 1, PCONV_KC1, false, 3>(ARGS);
  } else if (k == 1 && stride == 1 && !gate && !d2w && act != 4 && residual && pipe_way_out()) {
    // (the pipelined way out, see conv_epilogue_pipe; PCONV_CONV1X1_WAYOUT=batch: the generic one)
    if (cout > 96)
      rc = launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, false, 2>(ARGS);
    else if (cout > 32)
      rc = launch_conv<3, 1, 1, 8, 1, 1, PCONV_KC1, false, 2>(ARGS);
    else
      rc = launch_conv<1, 1, 1, 4, 1, 1, PCONV_KC1, false, 2>(ARGS);
  } else if (k == 1 && stride == 1) {
    BY_TILE(1, 1, PCONV_KC1)
  } else if (cout > 32 && !gate && !d2w && act != 4 && quad_way_out()) {
    // (1x1 stride 2, the shortcuts of the Down blocks: same tiles, same quad way out)
    if (cout > 96)
      rc = residual ? launch_conv<3, 1, 2, 4, 1, 2, PCONV_KC1, false, 4>(ARGS) : launch_conv<3, 1, 2, 4, 1, 2, PCONV_KC1, false, 3>(ARGS);
    else
      rc = residual ? launch_conv<3, 1, 1, 8, 1, 2, PCONV_KC1, false, 4>(ARGS) : launch_conv<3, 1, 1, 8, 1, 2, PCONV_KC1, false, 3>(ARGS);
  } else {
    BY_TILE(1, 2, PCONV_KC1)
  }
#undef BY_TILE
#undef ARGS
  if (rc != PCONV_OK) return rc;
  PCONV_LAUNCH_CHECK("conv2d");
  return PCONV_OK;
}

// GDN / inverse GDN of PseudoGDNV2.forward (PseudoContextV2.py:133-216) as ONE launch:
// norm = conv1x1(x^2, gamma) + beta on the matrix cores, then x / sqrt(norm) (or
// x * sqrt(norm)) in the epilogue, (+ residual,) zeros past each tile's valid width.
// The reference runs it as mask, square, conv, sqrt, three mask blends and a divide
// -- eight passes over the activation.
extern "C" int pconv_gdn(const float *in, const float *packed_gamma, const float *beta, float *out, int tn,
                         int ch, int h, int w, int inverse, const int32_t *col_limit, int npart,
                         const float *residual, const long long *views, void *stream) {
  PCONV_REQUIRE(in && packed_gamma && beta && out && in != out && residual != out, "gdn: bad pointer");
  PCONV_REQUIRE(tn > 0 && ch > 0 && h > 0 && w > 0, "gdn: bad shape");
  PCONV_REQUIRE(!col_limit || npart > 0, "gdn: col_limit needs npart");
  int cp, rp;
  pconv_conv_packed_size(ch, ch, 1, &cp, &rp);
  hipStream_t s = as_stream(stream);
  // views: in, out, residual
  const ConvView vin = view_at(views, 0, ch, h, w), vout = view_at(views, 1, ch, h, w);
  const ConvEpilogue ep = {beta, nullptr, residual, nullptr, col_limit, npart, inverse ? 3 : 2, 1,
                           view_at(views, 2, ch, h, w), dense_view(ch, h, w), 0};
  PCONV_REQUIRE(view_ok(vin, ch, h, w) && view_ok(vout, ch, h, w) && (!residual || view_ok(ep.vres, ch, h, w)),
                "gdn: strides overlap");
  int rc;
  if (use_resident_1x1(ch, ch, tn, h, w) && ch > 96 && residual)
    rc = launch_conv1x1<2, true, true>(in, packed_gamma, out, tn, ch, h, w, ch, cp, vin, vout, ep, s);
  else if (use_resident_1x1(ch, ch, tn, h, w) && ch > 96)
    rc = launch_conv1x1<2, true, false>(in, packed_gamma, out, tn, ch, h, w, ch, cp, vin, vout, ep, s);
  else if (use_resident_1x1(ch, ch, tn, h, w) && residual)
    rc = launch_conv1x1<1, true, true>(in, packed_gamma, out, tn, ch, h, w, ch, cp, vin, vout, ep, s);
  else if (use_resident_1x1(ch, ch, tn, h, w))
    rc = launch_conv1x1<1, true, false>(in, packed_gamma, out, tn, ch, h, w, ch, cp, vin, vout, ep, s);
  else if (quad_way_out() && ch > 96)
    rc = residual ? launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, true, 4>(in, packed_gamma, out, tn, ch, h, w, ch, cp, h, w, vin, vout, ep, s)
                  : launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, true, 3>(in, packed_gamma, out, tn, ch, h, w, ch, cp, h, w, vin, vout, ep, s);
  else if (pipe_way_out() && ch > 96 && residual)
    rc = launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, true, 2>(in, packed_gamma, out, tn, ch, h, w, ch, cp, h, w, vin, vout, ep, s);
  else if (pipe_way_out() && ch > 96)
    rc = launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, true, 1>(in, packed_gamma, out, tn, ch, h, w, ch, cp, h, w, vin, vout, ep, s);
  else if (ch > 96)
    rc = launch_conv<3, 1, 2, 4, 1, 1, PCONV_KC1, true>(in, packed_gamma, out, tn, ch, h, w, ch, cp, h, w, vin, vout, ep, s);
  else if (ch > 32)
    rc = launch_conv<3, 1, 1, 8, 1, 1, PCONV_KC1, true>(in, packed_gamma, out, tn, ch, h, w, ch, cp, h, w, vin, vout, ep, s);
  else
    rc = launch_conv<1, 1, 1, 4, 1, 1, PCONV_KC1, true>(in, packed_gamma, out, tn, ch, h, w, ch, cp, h, w, vin, vout, ep, s);
  if (rc != PCONV_OK) return rc;
  PCONV_LAUNCH_CHECK("gdn");
  return PCONV_OK;
}
