// ROUND-4 EXPERIMENT, NOT PART OF THE BUILD (kept for the record; results: DESIGN.md section 5,
// profiles/round4_band_*).  The entropy engine's kernels in the "band" form with the causal-compact
// reduction order: diagonals of windows staged in LDS by 16-byte LDS-DMA, weights in registers, only the
// unmasked entries enumerated.  Parity-green against the oracle (order 2) and the per-op kernel on the GPU
// (64 + 13 tests), measured, and NOT adopted: decode 102 / 110 / 141 / 175 ms for 1 / 2 / 4 / 8 frames
// (round 3: 94 / 110 / 141 / 178), entropy encode 39 / 57 / 88 / 152 ms (round 3: 29 / 40 / 59 / 98).
// This is csrc/entropy_engine.hip as of the experiment's last state (it needs ee_kernels.h / engine.cpp of
// commit "Entropy engine: band kernels ..." to build).
// Channels-last kernels of the native entropy engine (see ee_kernels.h).
//
// Same arithmetic as the per-op kernels of entropy.hip (the streams must be
// byte-identical), different memory layout: with [tile][row][col][C] storage a row of
// the 5 x 5 x C windows of neighbouring positions is ONE contiguous run, so the band
// kernels below stage whole diagonals of windows in LDS with kilobyte LDS-DMA pieces
// (rounds 1-3: 64-lane 4-byte gathers per window, bound by the L2 -> register path).
//
// Halos are written by the producer of the interior value they derive from (see
// ee_kernels.h): the causal rule of pconv_host_causal_table (what
// EntropyCtxPadRun2 stores in the per-op path, one launch per layer per step) is
// applied in the epilogue of the kernel that computes the value, so consumers
// read plain padded windows.
#include <stdlib.h>
#include <atomic>
#include "common.h"
#include "ee_kernels.h"
#include "gmm_device.h"

namespace {

constexpr int kWave = 64;
constexpr int K = 5, KK = 25, HALF = 2, PAD = 2, GO = 3;

struct Pos {
  int tw, row, tg, th;
};
__device__ __forceinline__ Pos decode_pos(int hw, int h, int w) {
  Pos p;
  p.tw = hw % w;
  p.row = hw / w;
  p.tg = p.row / h;
  p.th = p.row - p.tg * h;
  return p;
}

// The canonical butterfly of the masked convolution -- v[l] + v[l ^ off] for off = 32, 16, 8, 4, 2, 1: every
// lane ends with the same total, bit for bit what __shfl_xor gives (each step adds the same two numbers) --
// for 12 values at once (4 positions x 3 outputs), on the cross-lane VALU paths of gfx950 instead of
// ds_bpermute round trips: half / row swaps, a row rotate, one ds_swizzle (xor 4 has no DPP form), two quad
// permutes.  A plain butterfly repeats every exchange in both partners; here a step keeps each
// pair's sum in only one of them and uses the freed half for another value, so the 12
// reductions take 6 + 3 + 2 + 1 + 1 + 1 exchange-adds instead of 72.  Every value still
// goes through the pairs (l, l^32), (l, l^16), ... (l, l^1) in that order -- the sums
// are the same IEEE additions, bit for bit.  Result, per lane: the total of
// v[lane >> 4][{0, 2, 1, 2}[(lane >> 2) & 3]] (rows = positions, quads = outputs).
__device__ __forceinline__ float butterfly12(const float (&v)[4][GO], int lane) {
  // xor 32: pair value i (kept in lanes 0-31) with value i+6 (lanes 32-63)
  float a[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int i0 = k, i1 = k + 6;
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i0 / GO][i0 % GO]),
                                              __float_as_uint(v[i1 / GO][i1 % GO]), false, false);
    a[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  // xor 16: rows (16 lanes) 0..3 <- values k, k+3, k+6, k+9
  float b[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[k]), __float_as_uint(a[k + 3]), false, false);
    b[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  // xor 8 inside a row; lanes 0-7 of a row keep output 0, lanes 8-15 output 1; output 2 alone
  float c[3];
#pragma unroll
  for (int k = 0; k < 3; k++)
    c[k] = b[k] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b[k]), 0x128, 0xF, 0xF, false));
  const float c01 = (lane & 8) ? c[1] : c[0];
  // xor 4; lanes with bit 2 clear keep outputs 0 / 1, the others output 2
  const float d01 = c01 + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(c01), 0x101F));
  const float d2 = c[2] + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(c[2]), 0x101F));
  float t = (lane & 4) ? d2 : d01;
  t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x4E, 0xF, 0xF, false));  // xor 2
  t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xF, 0xF, false));  // xor 1
  return t;
}

// ---- the causal-compact reduction order (round 4) ---------------------------------------------------
//
// The causal mask lets output group tc use window entry (kh, kw, input channel ci) iff
//     ci / group_in + kh + kw < T,   T = tc + 4 + slack   (slack 0: input layer / constrain 5, 1: hidden / 6)
// (mask_constrain_cuda.cu:64-88; entropy_conv_cuda_v2.cu:326-380 evaluates the same rule as a channel limit
// per tap) -- on average HALF of the 25 x cin entries.  Until round 3 every kernel multiplied the masked half
// by zeros.  Now only the usable entries are enumerated, by window anti-diagonal d = kh + kw, then kh, then ci:
//     e = 0;  for d in 0..8:  U = clamp(T - d, 0, ngroup) * group_in
//               for kh in max(0, d-4) .. min(4, d):  for ci in 0..U-1:  entry e++ = (kh, d - kh, ci)
// lane e % 64 of a wave accumulates its entries in ascending e with fmaf from 0, then the xor butterfly
// 32..1, then + bias, PReLU, + residual.  This order is part of the bitstream contract: the step kernel, the
// bulk (encoder) kernel, the per-op kernel (entropy.hip) and the oracle (orc_entropy_conv, order 2) restate it.
// A group needs ceil(L / 64) rounds instead of ceil(25 cin / 64): 2 .. 16 instead of 17 for the hidden layers.
__host__ __device__ constexpr int slab_slots(int cin) { return (cin * KK + kWave - 1) / kWave * kWave; }
__host__ __device__ constexpr int slab_floats(int cin) { return slab_slots(cin) * 4; }

// number of usable entries of a group with threshold T
__host__ __device__ inline int compact_len(int T, int ngroup, int gin) {
  int L = 0;
  for (int d = 0; d <= 2 * (K - 1); d++) {
    int ug = T - d;
    ug = ug > ngroup ? ngroup : ug;
    if (ug <= 0) break;
    L += (d < K ? d + 1 : 2 * K - 1 - d) * ug * gin;
  }
  return L;
}

// Packed weights: for every (set, output group) one slab [slot e][4] in the compact order of the group:
// {w of the group's 3 outputs, info}, info = kh << 16 | (kw * cin + ci) (as int bits): where the entry sits
// in a window.  Slots past the group's L are {0, 0, 0, 0}: they multiply entry (0, 0, 0) by zero.
__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ packed, int cin, int ngroup,
                                   int slack, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int red = cin * KK, slots = (red + kWave - 1) / kWave * kWave;
  const int o = i & 3, grp = (i >> 2) / slots;  // grp = set*ngroup + tc
  int rem = (i >> 2) % slots;
  const int tc = grp % ngroup, gin = cin / ngroup, T = tc + 2 * HALF + slack;
  int kh = 0, kw = 0, ci = 0;
  bool live = false;
  for (int d = 0; d <= 2 * (K - 1) && !live; d++) {
    int ug = T - d;
    ug = ug > ngroup ? ngroup : ug;
    if (ug <= 0) break;
    const int U = ug * gin, kh0 = d < K ? 0 : d - (K - 1), ntap = d < K ? d + 1 : 2 * K - 1 - d;
    if (rem < ntap * U) {
      kh = kh0 + rem / U;
      ci = rem % U;
      kw = d - kh;
      live = true;
    } else {
      rem -= ntap * U;
    }
  }
  float v;
  if (o < GO)
    v = live ? w[((size_t)grp * GO + o) * red + ci * KK + kh * K + kw] : 0.f;
  else
    v = __int_as_float(live ? (kh << 16) | (kw * cin + ci) : 0);
  packed[i] = v;
}

// ---- halos ---------------------------------------------------------------

// element index of padded (tile, row, col) inside one image of C channels
__device__ __forceinline__ size_t tile_elem(int tile, int prow, int pcol, int h, int w, int C) {
  return (((size_t)tile * (h + 2 * PAD) + prow) * (w + 2 * PAD) + pcol) * C;
}

// Rewrites halo entry `en` (index into the dense causal table: tile, side, halo
// row, column) of channel ch from the current values of its two source columns;
// SUBST: the caller has just produced the source at (own_row, own_col) and passes
// its value in a register instead of re-reading its own store.
template <bool SUBST>
__device__ __forceinline__ void halo_write(const EeGeom &g, float *img, int C, int ch, int en, int own_row,
                                           int own_col, float own_val) {
  const int h = g.h, w = g.w;
  const int cp = en % w;
  int q = en / w;
  const int r = q % PAD;
  q /= PAD;
  const int side = q & 1, tg = q >> 1;
  const int c = g.vh_col[en];
  if (c == -2) return;  // no causal source
  const int srow = side ? (tg + 1) * h + r : tg * h - PAD + r;
  if (srow < 0 || srow >= h * g.npart) return;
  const int st = srow / h, sr = srow - st * h;
  const int wst = g.widths[st];
  int c1 = c + 1;
  c1 = c1 >= wst ? c1 - wst : c1;
  const float t = g.vh_wgt[en];
  float a = 0.f, b;
  if (c >= 0)
    a = (SUBST && srow == own_row && c == own_col) ? own_val : img[tile_elem(st, sr + PAD, c + PAD, h, w, C) + ch];
  b = (SUBST && srow == own_row && c1 == own_col) ? own_val : img[tile_elem(st, sr + PAD, c1 + PAD, h, w, C) + ch];
  const float v = a * t + b * (1 - t);
  const size_t dst = tile_elem(tg, side ? h + PAD + r : r, cp + PAD, h, w, C) + ch;
  img[dst] = v;
  if (cp < PAD) img[dst + (size_t)g.widths[tg] * C] = v;  // circular wrap of the first columns
}

// bulk: every halo entry and every wrap column of `nrep` images from the interior
__global__ void ee_halo_bulk_kernel(EeGeom g, float *__restrict__ buf, int C, long long n_halo, long long n_wrap) {
  const int h = g.h, w = g.w;
  const size_t img_elems = (size_t)g.npart * (h + 2 * PAD) * (w + 2 * PAD) * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_halo + n_wrap;
       i += (long long)gridDim.x * blockDim.x) {
    if (i < n_halo) {
      const int ch = (int)(i % C);
      const long long q = i / C;
      const int entries = g.npart * 2 * PAD * w;
      const int en = (int)(q % entries);
      const int rep = (int)(q / entries);
      const int tg = en / w / PAD / 2;
      if (en % w >= g.widths[tg]) continue;  // dead column: never read
      halo_write<false>(g, buf + rep * img_elems, C, ch, en, -1, -1, 0.f);
    } else {
      const long long j = i - n_halo;
      const int ch = (int)(j % C);
      long long q = j / C;
      const int k = (int)(q % PAD);
      q /= PAD;
      const int th = (int)(q % h);
      q /= h;
      const int tile = (int)(q % g.npart);
      const int rep = (int)(q / g.npart);
      float *img = buf + rep * img_elems;
      const size_t src = tile_elem(tile, th + PAD, k + PAD, h, w, C) + ch;
      img[src + (size_t)g.widths[tile] * C] = img[src];
    }
  }
}

// ---- layers --------------------------------------------------------------
//
// One design for the decoder's step and the encoder's bulk pass ("band kernels", round 4).
//
// Geometry.  Plane ps of the wavefront crosses latitude tile t in the positions (th, tw = P - th), P = ps - t*h:
// an anti-diagonal.  The 5 x 5 windows of NB consecutive rows of that diagonal lie in a BAND of NB + 4 rows
// x 9 columns of the (padded) tile, and band column kh + kw is exactly the window anti-diagonal d of the
// causal rule.  A workgroup = NPL waves takes NB rows of NPL consecutive planes of one tile of one (set,
// image): their bands overlap, shifted by one column per plane, so the union -- (NB + 4) rows x (8 + NPL)
// columns x C channels, every row a contiguous run of the channels-last buffer -- is staged ONCE in LDS with
// 16-byte LDS-DMA (coalesced kilobyte pieces instead of 64-lane 4-byte gathers per window: the step kernel
// of rounds 1-3 pulled every window out of L2 separately, 17 gathers per position, and was bound by them).
// Window entry (kh, kw, ci) of the position in band row p of plane k sits at
//     band[(p + kh) * S + (kh + kw + k) * C + ci] = p * S + k * C + [kh * (S + C) + kw * C + ci]:
// a position- and plane-independent offset per entry, packed with the weights (`info`).
//
// Wave k owns plane k: output group tc = psum - plane.  Its weights -- the group's slab in causal-compact
// order, only ceil(L / 64) rounds of it -- go to REGISTERS once and serve the 8 positions of a pass; per
// position and round a lane issues one ds_read_b32 (base + immediate) and three fmaf, the reads of round
// i + 1 are in flight under the fmaf of round i; the 24 sums leave through two packed butterflies.
//   step (decoder):  grid (tiles x row chunks, plane chunks of the step's window, 3 sets x images); weights
//                    straight from global memory (every wave another group); what the way out reads
//                    (residual, halo records) is requested before the rounds; it writes the value, its
//                    circular-wrap copy and the halo entries interpolated from it.
//   bulk (encoder):  all planes; a wave walks ALL groups of its plane's positions, the group's slab staged in
//                    LDS for the workgroup (double buffered, one barrier per group); the outputs of the
//                    workgroup's positions collect in an LDS tile (initialised with the residual by LDS-DMA)
//                    and leave once, as runs of 3 G floats per position -- no global-memory instruction in
//                    the group loop but the slab DMA, so its barrier waits for nothing else; halos by
//                    ee_halo_bulk.
// Same device functions, same per-output operations in the same order: encoder and decoder tables agree bit
// for bit, and both equal the per-op kernel and the oracle (order 2).
template <int CIN, int NPL_, int NB_, bool BULK_>
struct Band {
  static constexpr int C = CIN, NPL = NPL_, NB = NB_;
  static constexpr bool BULK = BULK_;
  static constexpr int PB = 8;                      // positions per accumulation pass
  static constexpr int NPASS = NB / PB;
  static constexpr int ND = 8 + NPL;                // band columns
  static constexpr int RL = ND * C;                 // floats of a band row
  static constexpr int S = (RL + 3) / 4 * 4;        // row stride: rows start on 16-byte boundaries
  static constexpr int NBR = NB + 2 * PAD;          // band rows
  static constexpr int A = S + C;                   // address step of kh
  static constexpr int ITER = slab_slots(CIN) / kWave;
  static constexpr int NCH = (ITER + 19) / 20;      // weight rounds kept in registers at a time: <= 20
  static constexpr int WCH = (ITER + NCH - 1) / NCH;
  static constexpr int BLOCK = NPL * kWave;
  static constexpr int PIECES = (RL + 255) / 256;   // 16-byte DMA instructions per band row
  static constexpr int BAND_FLOATS = NBR * S;
  // bulk: the group's slab staged in LDS for the workgroup (where two of them fit beside the band: cin <= 48;
  // the wide models' waves fetch their rounds from global memory as the step kernel does)
  static constexpr bool SLAB = BULK && CIN <= 48;
  static constexpr int SLAB_FLOATS = SLAB ? 2 * ITER * kWave * 4 : 0;
  static constexpr int TAB_FLOATS = 2 * GO * CIN;   // bias / slope of a set (cout = 3 ngroup <= 3 CIN)
  static constexpr int YT_FLOATS = BULK ? NPL * NB * GO * CIN : 0;  // bulk: output tile [plane][row][cout]
  static constexpr size_t LDS_BYTES = (size_t)(BAND_FLOATS + SLAB_FLOATS + TAB_FLOATS + YT_FLOATS) * 4;
  static_assert(NB % PB == 0 && RL % 4 == 0, "band geometry");
  static_assert(((NB - 1) * S + NPL * C + 4 * A + 4 * C + C) * 4 < (1 << 16) * 4, "LDS offsets");
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef const __attribute__((address_space(3))) float lds_float_t;

// 8 positions x 3 outputs: acc[p][o] += sum over the wave's rounds of x(entry, position p) * w(entry, o).
// xa[i]: LDS byte address of the lane's entry of round i for position 0 of the pass (slots past the rounds:
// any valid address); position p is p * S floats further (an immediate).  n >= 1: rounds (uniform).  The
// reads of round i + 1 are issued before the fmaf of round i (LDS returns in order: counted waits).
template <class B, int I>
__device__ __forceinline__ void band_round(float (&acc)[B::PB][GO], const float4 (&wv)[B::WCH], const unsigned (&xa)[B::WCH],
                                           int n, float (&xcur)[B::PB]) {
  float xnext[B::PB];
  if constexpr (I + 1 < B::WCH) {
#pragma unroll
    for (int p = 0; p < B::PB; p++) xnext[p] = *(lds_float_t *)(uintptr_t)(xa[I + 1] + (unsigned)(p * B::S * 4));
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int p = 0; p < B::PB; p++) {
    acc[p][0] = fmaf(xcur[p], wv[I].x, acc[p][0]);
    acc[p][1] = fmaf(xcur[p], wv[I].y, acc[p][1]);
    acc[p][2] = fmaf(xcur[p], wv[I].z, acc[p][2]);
  }
  if constexpr (I + 1 < B::WCH) {
    // (pins the reads of round I + 1 in front of this round's fmaf: left alone the compiler sinks them into
    // the next round's block -- or behind the fmaf -- and every round starts with an exposed LDS round trip)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < B::PB; p++) asm volatile("" : "+v"(xnext[p]));
    if (I + 1 < n) band_round<B, I + 1>(acc, wv, xa, n, xnext);  // (uniform)
  }
}
template <class B>
__device__ __forceinline__ void band_rounds(float (&acc)[B::PB][GO], const float4 (&wv)[B::WCH], const unsigned (&xa)[B::WCH],
                                            int n) {
  float x0[B::PB];
#pragma unroll
  for (int p = 0; p < B::PB; p++) x0[p] = *(lds_float_t *)(uintptr_t)(xa[0] + (unsigned)(p * B::S * 4));
  band_round<B, 0>(acc, wv, xa, n, x0);
}

// info of a packed entry -> LDS byte address of the entry for band row `row0`, plane k
template <class B>
__device__ __forceinline__ unsigned band_entry_addr(float info, unsigned band_base, int k, int row0) {
  const unsigned u = (unsigned)__float_as_int(info);
  return band_base + 4u * ((u >> 16) * (unsigned)B::A + (u & 0xffffu) + (unsigned)(k * B::C + row0 * B::S));
}

// The two packed butterflies of a pass: lane L gets, for half = 0 / 1, the total of position 4 half + (L >> 4),
// output {0, 2, 1, 2}[(L >> 2) & 3].
template <class B>
__device__ __forceinline__ void band_totals(const float (&acc)[B::PB][GO], int lane, float (&tot)[2]) {
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const float a4[4][GO] = {{acc[4 * half][0], acc[4 * half][1], acc[4 * half][2]},
                             {acc[4 * half + 1][0], acc[4 * half + 1][1], acc[4 * half + 1][2]},
                             {acc[4 * half + 2][0], acc[4 * half + 2][1], acc[4 * half + 2][2]},
                             {acc[4 * half + 3][0], acc[4 * half + 3][1], acc[4 * half + 3][2]}};
    tot[half] = butterfly12(a4, lane);
  }
}

// ---- step kernel (decoder) ------------------------------------------------------------------------------
template <int CIN, int NPL, int NB>
__global__ __launch_bounds__(NPL * kWave) void ee_band_step_kernel(EeGeom g, const float *__restrict__ x, int shared_input,
                                                                   const float *__restrict__ wp,
                                                                   const float *__restrict__ bias,
                                                                   const float *__restrict__ slope,
                                                                   const float *__restrict__ residual,
                                                                   float *__restrict__ y, int pad_out, int slack,
                                                                   int first_plane, int nplane, int psum) {
  typedef Band<CIN, NPL, NB, false> B;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *band = lds;
  const int h = g.h, w = g.w, win = w + 2 * PAD;
  const int cout = GO * g.ngroup, gin = CIN / g.ngroup;
  const int row_chunks = (h + NB - 1) / NB;
  const int tile = blockIdx.x / row_chunks;
  const int th_lo = (blockIdx.x - tile * row_chunks) * NB;
  const int plane0 = first_plane + blockIdx.y * NPL;       // plane of wave 0
  const int pn = blockIdx.z;                               // replica-major image index: set * nimg + img
  const int set = (pn >= g.nimg) + (pn >= 2 * g.nimg);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int width = g.widths[tile];
  const int nrows = h - th_lo < NB ? h - th_lo : NB;
  const int P0 = plane0 - tile * h;                        // column of wave 0's diagonal in row 0 of the tile
  // does any position (th_lo + p, P0 + k - th_lo - p), p < nrows, k < planes here, exist?
  const int nplanes_here = first_plane + nplane - plane0 < NPL ? first_plane + nplane - plane0 : NPL;
  if (nplanes_here <= 0 || P0 + nplanes_here - 1 - th_lo < 0 || P0 - th_lo - (nrows - 1) >= width) return;  // (uniform)

  // ---- stage the band: rows th_lo .. th_lo + nrows + 3 (padded coordinates) of the tile, row b from padded
  // column P0 - th_lo - b on, ND columns.  A row is one contiguous run of the channels-last buffer (it may
  // start left of the tile row or end right of it: those bytes belong to columns no existing position reads,
  // and the engine allocates its buffers with a guard band so that the addresses are valid).
  const size_t in_img = (size_t)g.npart * (h + 2 * PAD) * win * CIN;
  const float *ximg = x + (size_t)(shared_input ? pn - set * g.nimg : pn) * in_img;
  {
    const long long row0 = ((long long)tile * (h + 2 * PAD) + th_lo) * win + (P0 - th_lo);
    const int nbr = nrows + 2 * PAD;
    for (int b = wave; b < nbr; b += NPL) {
      const float *src = ximg + (row0 + (long long)b * (win - 1)) * CIN;
#pragma unroll
      for (int j = 0; j < B::PIECES; j++) {
        const int f = (j * kWave + lane) * 4;
        if (f < B::RL)
          __builtin_amdgcn_global_load_lds((glb_void_t *)(src + f), (lds_void_t *)(band + b * B::S + j * 256), 16, 0, 0);
      }
    }
  }
  const int plane = plane0 + wave;
  const int P = P0 + wave;
  // rows of this wave's diagonal that exist: th in [th_lo, th_lo + nrows), 0 <= P - th < width
  const bool wave_live = wave < nplanes_here && P - th_lo >= 0 && P - th_lo - (nrows - 1) < width;
  const size_t out_img = (size_t)g.npart * (h + 2 * pad_out) * (w + 2 * pad_out) * cout;
  const float *rimg = residual ? residual + (size_t)pn * out_img : nullptr;
  float *yimg = y + (size_t)pn * out_img;
  const unsigned band_base = (unsigned)(uintptr_t)band;  // low half of the flat address = LDS byte offset
  const int tc = wave_live ? psum - plane : 0;
  const int L = compact_len(tc + 2 * HALF + slack, g.ngroup, gin);
  const int niter = wave_live ? (L + kWave - 1) / kWave : 0;
  const float4 *wg = reinterpret_cast<const float4 *>(wp) + ((size_t)set * g.ngroup + tc) * (B::ITER * kWave) + lane;
  float4 wv[B::WCH];
  unsigned xa[B::WCH];
  // the first (for cin <= 48: the only) chunk of weight rounds is requested before the band has landed
#pragma unroll
  for (int i = 0; i < B::WCH; i++) {
    wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < niter) wv[i] = wg[i * kWave];
  }
  // the lane's part in the way out: output `out` of position row (lane >> 4) + 4 half of a pass
  const int quad = (lane >> 2) & 3;
  const int out = quad == 0 ? 0 : (quad == 2 ? 1 : 2);
  const int ch = tc * GO + out;
  const float bv = bias[set * cout + ch];
  const float sv = slope ? slope[set * cout + ch] : 1.f;  // (v * 1 is v)
  __syncthreads();  // the band is in LDS (the barrier waits for vmcnt(0): the DMA and the loads above)
  if (!wave_live) return;
#pragma unroll 1
  for (int pass = 0; pass < B::NPASS; pass++) {
    const int th0 = th_lo + pass * B::PB;
    if (pass * B::PB >= nrows || P - th0 < 0 || P - th0 - (B::PB - 1) >= width) continue;  // (uniform) nothing in this pass
    // what the way out will read, requested now: the residual and, for edge rows, the reverse-halo records
    bool live[2];
    size_t pix[2];
    float rv[2];
    int rev[2];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int th = th0 + 4 * half + (lane >> 4), tw = P - th;
      live[half] = quad != 3 && th < h && tw >= 0 && tw < width;
      pix[half] = pad_out ? ((size_t)tile * (h + 2 * PAD) + th + PAD) * win + tw + PAD : ((size_t)tile * h + th) * w + tw;
      rv[half] = 0.f;
      rev[half] = 0;
      if (live[half]) {
        if (rimg) rv[half] = rimg[pix[half] * cout + ch];
        if (pad_out && (th < PAD || th >= h - PAD)) rev[half] = g.pix_rev[((size_t)tile * h + th) * w + tw];
      }
    }
    float acc[B::PB][GO];
#pragma unroll
    for (int p = 0; p < B::PB; p++) acc[p][0] = acc[p][1] = acc[p][2] = 0.f;
#pragma unroll 1
    for (int c = 0; c < B::NCH; c++) {
      const int n = niter - c * B::WCH < B::WCH ? niter - c * B::WCH : B::WCH;
      if (n <= 0) break;
      if (c > 0 || pass > 0) {
#pragma unroll
        for (int i = 0; i < B::WCH; i++)
          if (i < n) wv[i] = wg[(c * B::WCH + i) * kWave];
      }
#pragma unroll
      for (int i = 0; i < B::WCH; i++) xa[i] = band_entry_addr<B>(wv[i].w, band_base, wave, pass * B::PB);  // (slots past n: info 0, a valid address)
      band_rounds<B>(acc, wv, xa, n);
    }
    float tot[2];
    band_totals<B>(acc, lane, tot);
    // way out: bias, PReLU, residual, the value, its circular-wrap copy, the halo entries interpolated from it
    // (halo_write; the four lanes of a quad hold the same value and share its records)
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (!live[half]) continue;
      const int tw = P - (th0 + 4 * half + (lane >> 4));
      float v = tot[half] + bv;
      if (v < 0) v = v * sv;
      if (rimg) v = v + rv[half];
      const bool writer = (lane & 3) == 0;
      if (writer) yimg[pix[half] * cout + ch] = v;
      if (pad_out) {
        if (tw < PAD && writer) yimg[(pix[half] + width) * cout + ch] = v;  // circular wrap copy
        const int nrev = rev[half] & 15;
        const EeHalo *hr = g.halo + (rev[half] >> 4);
        for (int k = lane & 3; k < nrev; k += 4) {
          const EeHalo q = hr[k];
          const float other = (q.info & (1 << 29)) ? v : (q.other >= 0 ? yimg[(size_t)q.other * cout + ch] : 0.f);
          const float a = (q.info & (1 << 30)) ? other : v, b = (q.info & (1 << 30)) ? v : other;
          const float hv = a * q.t + b * (1 - q.t);
          float *dst = yimg + (size_t)q.dst * cout + ch;
          *dst = hv;
          const int wd = q.info & 0xffff;
          if (wd) dst[(size_t)wd * cout] = hv;  // circular wrap of the first columns
        }
      }
    }
  }
}

// ---- bulk kernel (encoder) ------------------------------------------------------------------------------
template <int CIN, int NPL, int NB>
__global__ __launch_bounds__(NPL * kWave) void ee_band_bulk_kernel(EeGeom g, const float *__restrict__ x, int shared_input,
                                                                   const float *__restrict__ wp,
                                                                   const float *__restrict__ bias,
                                                                   const float *__restrict__ slope,
                                                                   const float *__restrict__ residual,
                                                                   float *__restrict__ y, int pad_out, int slack,
                                                                   int nplane) {
  typedef Band<CIN, NPL, NB, true> B;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *band = lds;
  float *slab = lds + B::BAND_FLOATS;                                    // two group slabs
  float *tab = lds + B::BAND_FLOATS + B::SLAB_FLOATS;                    // bias [cout], slope [cout]
  float *ytile = lds + B::BAND_FLOATS + B::SLAB_FLOATS + B::TAB_FLOATS;  // [plane k][row p][cout]
  const int h = g.h, w = g.w, win = w + 2 * PAD;
  const int cout = GO * g.ngroup, gin = CIN / g.ngroup;
  const int row_chunks = (h + NB - 1) / NB;
  const int tile = blockIdx.x / row_chunks;
  const int th_lo = (blockIdx.x - tile * row_chunks) * NB;
  const int plane0 = blockIdx.y * NPL;
  const int pn = blockIdx.z;
  const int set = (pn >= g.nimg) + (pn >= 2 * g.nimg);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int width = g.widths[tile];
  const int nrows = h - th_lo < NB ? h - th_lo : NB;
  const int P0 = plane0 - tile * h;
  const int nplanes_here = nplane - plane0 < NPL ? nplane - plane0 : NPL;
  if (nplanes_here <= 0 || P0 + nplanes_here - 1 - th_lo < 0 || P0 - th_lo - (nrows - 1) >= width) return;  // (uniform)
  const size_t in_img = (size_t)g.npart * (h + 2 * PAD) * win * CIN;
  const float *ximg = x + (size_t)(shared_input ? pn - set * g.nimg : pn) * in_img;
  {
    const long long row0 = ((long long)tile * (h + 2 * PAD) + th_lo) * win + (P0 - th_lo);
    const int nbr = nrows + 2 * PAD;
    for (int b = wave; b < nbr; b += NPL) {
      const float *src = ximg + (row0 + (long long)b * (win - 1)) * CIN;
#pragma unroll
      for (int j = 0; j < B::PIECES; j++) {
        const int f = (j * kWave + lane) * 4;
        if (f < B::RL)
          __builtin_amdgcn_global_load_lds((glb_void_t *)(src + f), (lds_void_t *)(band + b * B::S + j * 256), 16, 0, 0);
      }
    }
  }
  const size_t out_img = (size_t)g.npart * (h + 2 * pad_out) * (w + 2 * pad_out) * cout;
  const float *rimg = residual ? residual + (size_t)pn * out_img : nullptr;
  float *yimg = y + (size_t)pn * out_img;
  // pixel (in y's layout) of position (plane k, row p); -1: no such position
  auto out_pixel = [&](int k, int p) -> long long {
    const int th = th_lo + p, tw = P0 + k - th;
    if (k >= nplanes_here || th >= h || tw < 0 || tw >= width) return -1;
    return pad_out ? ((long long)tile * (h + 2 * PAD) + th + PAD) * win + tw + PAD : ((long long)tile * h + th) * w + tw;
  };
  // the output tile starts as the residual (zeros where there is none), by 4-byte LDS-DMA: one instruction
  // fills 64 consecutive floats of ytile = [position][cout]
  const int ytn = NPL * NB * cout;
  for (int i0 = wave * kWave; i0 < ytn; i0 += B::BLOCK) {
    const int i = i0 + lane;
    const int pos = i / cout, c = i - pos * cout;
    const long long px = i < ytn ? out_pixel(pos / NB, pos % NB) : -1;
    if (rimg && px >= 0)
      __builtin_amdgcn_global_load_lds((glb_void_t *)(rimg + px * cout + c), (lds_void_t *)(ytile + i0), 4, 0, 0);
    else if (i < ytn)
      ytile[i] = 0.f;
  }
  for (int i = threadIdx.x; i < cout; i += B::BLOCK) {
    tab[i] = bias[set * cout + i];
    tab[cout + i] = slope ? slope[set * cout + i] : 1.f;  // (v * 1 is v)
  }
  const int P = P0 + wave;
  const bool wave_live = wave < nplanes_here && P - th_lo >= 0 && P - th_lo - (nrows - 1) < width;
  const unsigned band_base = (unsigned)(uintptr_t)band;
  const float4 *wset = reinterpret_cast<const float4 *>(wp) + (size_t)set * g.ngroup * (B::ITER * kWave);
  const int quad = (lane >> 2) & 3;
  const int out = quad == 0 ? 0 : (quad == 2 ? 1 : 2);
  // SLAB: the group's slab (only its ceil(L / 64) rounds) is staged for the workgroup by LDS-DMA, double
  // buffered: slab tc + 1 lands while group tc is computed.
  auto stage_slab = [&](int tc) {
    if constexpr (B::SLAB) {
      const int L = compact_len(tc + 2 * HALF + slack, g.ngroup, gin);
      const int n = (L + kWave - 1) / kWave;
      const float4 *src = wset + (size_t)tc * (B::ITER * kWave) + lane;
      float *dst = slab + (tc & 1) * (B::ITER * kWave * 4);
      for (int i = wave; i < n; i += NPL)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(src + i * kWave), (lds_void_t *)(dst + i * kWave * 4), 16, 0, 0);
    }
  };
  stage_slab(0);
  if constexpr (!B::SLAB) __syncthreads();  // band, table and output tile
#pragma unroll 1
  for (int tc = 0; tc < g.ngroup; tc++) {
    if constexpr (B::SLAB) {
      __syncthreads();  // slab tc (and, the first time, band, table, output tile) landed; everybody is done with slab tc - 1
      if (tc + 1 < g.ngroup) stage_slab(tc + 1);
    }
    if (!wave_live) continue;
    const int L = compact_len(tc + 2 * HALF + slack, g.ngroup, gin);
    const int niter = (L + kWave - 1) / kWave;
    const float4 *wg = wset + (size_t)tc * (B::ITER * kWave) + lane;
    const float4 *ws = reinterpret_cast<const float4 *>(slab + (tc & 1) * (B::ITER * kWave * 4)) + lane;
    const int ch = tc * GO + out;
    const float bv = tab[ch], sv = tab[cout + ch];
#pragma unroll 1
    for (int pass = 0; pass < B::NPASS; pass++) {
      const int th0 = th_lo + pass * B::PB;
      if (pass * B::PB >= nrows || P - th0 < 0 || P - th0 - (B::PB - 1) >= width) continue;
      float acc[B::PB][GO];
#pragma unroll
      for (int p = 0; p < B::PB; p++) acc[p][0] = acc[p][1] = acc[p][2] = 0.f;
#pragma unroll 1
      for (int c = 0; c < B::NCH; c++) {
        const int n = niter - c * B::WCH < B::WCH ? niter - c * B::WCH : B::WCH;
        if (n <= 0) break;
        float4 wv[B::WCH];
        unsigned xa[B::WCH];
#pragma unroll
        for (int i = 0; i < B::WCH; i++) {
          wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i < n) wv[i] = B::SLAB ? ws[(c * B::WCH + i) * kWave] : wg[(c * B::WCH + i) * kWave];
        }
#pragma unroll
        for (int i = 0; i < B::WCH; i++) xa[i] = band_entry_addr<B>(wv[i].w, band_base, wave, pass * B::PB);  // (slots past n: info 0, a valid address)
        band_rounds<B>(acc, wv, xa, n);
      }
      float tot[2];
      band_totals<B>(acc, lane, tot);
#pragma unroll
      for (int half = 0; half < 2; half++) {
        if (quad == 3 || (lane & 3)) continue;
        const int p = pass * B::PB + 4 * half + (lane >> 4);
        float v = tot[half] + bv;
        if (v < 0) v = v * sv;
        float *slot = ytile + (wave * NB + p) * cout + ch;
        *slot = v + *slot;  // (+ residual; garbage of positions that do not exist is never stored)
      }
    }
  }
  __syncthreads();
  // the workgroup's outputs leave once: per position 3 G consecutive floats of the channels-last buffer
  for (int i = threadIdx.x; i < ytn; i += B::BLOCK) {
    const int pos = i / cout, c = i - pos * cout;
    const long long px = out_pixel(pos / NB, pos % NB);
    if (px >= 0) yimg[px * cout + c] = ytile[i];
  }
}

// Decoder chain flags: three int32 in pinned (coherent) host memory per group, so that the
// host never launches on the critical path of a step -- the whole chain of a decode is
// queued ahead and its two ends talk through memory (engine.cpp):
//   flags[0]  host -> GPU   s+1 once the symbols of step s are in packed_h
//   flags[1]  GPU -> host   s+1 once the CDF rows of step s are in tables_h
//   flags[2]  GPU -> host   a scatter kernel gave up waiting (bounded spin)
// Only block (0, 0) polls the host flag (every poll is a PCIe round trip; dozens of pollers
// slow the fabric down for everybody); it passes the value on through a device word the
// other blocks of the launch poll in L2.
__device__ __forceinline__ void chain_wait(volatile int32_t *flags, int32_t *relay, int wait_for) {
  if (threadIdx.x == 0) {
    const bool leader = blockIdx.x == 0 && blockIdx.y == 0;
    int32_t *word = leader ? const_cast<int32_t *>(flags) : relay;
    const long long t0 = wall_clock64();  // 100 MHz
    for (;;) {
      const int seen = leader ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                              : __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= wait_for) break;
      if (leader)
        __builtin_amdgcn_s_sleep(16);
      else
        __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > 600000000LL) {  // 6 s: the host side is gone; every wave still exits
        __hip_atomic_store(const_cast<int32_t *>(flags) + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
    if (leader) __hip_atomic_store(relay, wait_for, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

// decoder: one thread per (image, position) of the step that was just decoded; grid.y = image.
// flags != null: first wait until the host has published the symbols (flags[0] >= wait_for).
__global__ void ee_scatter_kernel(EeGeom g, const float *__restrict__ packed, float *__restrict__ ctx, int lo,
                                  int len, int psum, float bias, volatile int32_t *flags, int32_t *relay,
                                  int wait_for) {
  if (flags) chain_wait(flags, relay, wait_for);
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= len) return;
  const int n = blockIdx.y;
  const int C = g.ngroup, win = g.w + 2 * PAD;
  const EePos rec = g.pos[lo + l];
  const int tc = psum - g.pos_plane[lo + l];
  float *img = ctx + (size_t)n * g.npart * (g.h + 2 * PAD) * win * C;
  const float v = packed[(size_t)n * len + l] + bias;
  float *dst = img + (size_t)(rec.pix + 2 * win + 2) * C + tc;
  *dst = v;
  if (rec.wrap) dst[(size_t)rec.wrap * C] = v;
  const int nrev = rec.rev & 15;
  const EeHalo *hr = g.halo + (rec.rev >> 4);
  for (int k = 0; k < nrev; k++) {
    const EeHalo q = hr[k];
    const float other = (q.info & (1 << 29)) ? v : (q.other >= 0 ? img[(size_t)q.other * C + tc] : 0.f);
    const float a = (q.info & (1 << 30)) ? other : v, b = (q.info & (1 << 30)) ? v : other;
    const float hv = a * q.t + b * (1 - q.t);
    float *hd = img + (size_t)q.dst * C + tc;
    *hd = hv;
    const int wd = q.info & 0xffff;
    if (wd) hd[(size_t)wd * C] = hv;
  }
}

// one thread per NCHW element of the symbol tensor
__global__ void ee_fill_ctx_kernel(EeGeom g, const float *__restrict__ sym, float *__restrict__ ctx, float bias,
                                   long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tw = (int)(i % g.w);
    const int th = (int)((i / g.w) % g.h);
    const int c = (int)((i / g.w / g.h) % g.ngroup);
    const long long tb = i / g.w / g.h / g.ngroup;  // image*npart + tile
    if (tw >= g.widths[tb % g.npart]) continue;
    ctx[(((size_t)tb * (g.h + 2 * PAD) + th + PAD) * (g.w + 2 * PAD) + tw + PAD) * g.ngroup + c] = sym[i] + bias;
  }
}

__global__ void ee_read_symbols_kernel(EeGeom g, const float *__restrict__ ctx, float *__restrict__ sym, float bias,
                                       long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tw = (int)(i % g.w);
    const int th = (int)((i / g.w) % g.h);
    const int c = (int)((i / g.w / g.h) % g.ngroup);
    const long long tb = i / g.w / g.h / g.ngroup;
    float v = 0.f;
    if (tw < g.widths[tb % g.npart])
      v = ctx[(((size_t)tb * (g.h + 2 * PAD) + th + PAD) * (g.w + 2 * PAD) + tw + PAD) * g.ngroup + c] + bias;
    sym[i] = v;
  }
}

// flags != null: the block that finishes last publishes flags[1] = publish (system scope)
// after every block's rows are out; `counter` (device, zero) counts the finished blocks.
__global__ void ee_tables_kernel(EeGeom g, const float *__restrict__ y, const float *__restrict__ symbols,
                                 int32_t *__restrict__ table, int32_t *__restrict__ labels, int lo, int len,
                                 int psum, int nstep, float bias, float total, float beta, int32_t *counter,
                                 volatile int32_t *flags, int publish) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < len) {
    const int n = blockIdx.y;
    const size_t r = (size_t)n * len + l;
    const int hw = g.pos[lo + l].hw;
    const int tc = psum - g.pos_plane[lo + l];
    const int cout = g.ngroup * 3;
    const size_t plane_px = (size_t)g.npart * g.h * g.w;
    float par[3][3];
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {
      const float *base = y + ((size_t)(rep * g.nimg + n) * plane_px + hw) * cout + tc * 3;
#pragma unroll
      for (int k = 0; k < 3; k++) par[rep][k] = base[k];
    }
    gmm_prepare_row(par[0], par[1], 3, beta);
    gmm_cdf_row<int32_t>(par[0], par[1], par[2], 3, nstep, bias, total, 1, table + r * (nstep + 1));
    if (symbols) {
      // NCHW symbol tensor: (image*npart + tile, group, row, col)
      const int hwt = g.h * g.w, tg = hw / hwt, inner = hw - tg * hwt;
      labels[r] = (int32_t)symbols[(((size_t)n * g.npart + tg) * g.ngroup + tc) * hwt + inner];
    }
  }
  if (flags) {
    __syncthreads();  // every wave of the block has drained its stores (vmcnt(0) before the barrier)
    if (threadIdx.x == 0) {
      __threadfence_system();
      const int nblocks = gridDim.x * gridDim.y;
      if (atomicAdd(counter, 1) == nblocks - 1) {
        *counter = 0;  // ready for the next step of this stream
        __threadfence_system();
        __hip_atomic_store(const_cast<int32_t *>(flags) + 1, publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// The same rows with EIGHT lanes per row (nstep == 8): lane q of an octet evaluates CDF entry
// q + 1 (the 3-gaussian sum of the one-thread form, same operations in the same order), the
// octet exchanges its 8 raw entries and every lane runs the short monotonicity repair on
// them, then stores its own entry.  The one-thread form spends ~21 erf evaluations per thread
// in a launch of a few hundred waves: pure latency, 3x longer than a layer of the network.
__global__ void ee_tables8_kernel(EeGeom g, const float *__restrict__ y, int32_t *__restrict__ table, int lo,
                                  int len, int psum, float bias, float total, float beta, int32_t *counter,
                                  volatile int32_t *flags, int publish) {
  constexpr int NS = 8;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = tid >> 3, q = tid & 7;
  const bool live = l < len;
  const int n = blockIdx.y;
  float cur = 0.f;
  if (live) {
    const int hw = g.pos[lo + l].hw;
    const int tc = psum - g.pos_plane[lo + l];
    const int cout = g.ngroup * 3;
    const size_t plane_px = (size_t)g.npart * g.h * g.w;
    float par[3][3];
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {
      const float *base = y + ((size_t)(rep * g.nimg + n) * plane_px + hw) * cout + tc * 3;
#pragma unroll
      for (int k = 0; k < 3; k++) par[rep][k] = base[k];
    }
    gmm_prepare_row(par[0], par[1], 3, beta);
    cur = gmm_cdf_entry(par[0], par[1], par[2], 3, NS, q + 1, bias, total);
  }
  // raw entries 1..8 of the row, from the 8 lanes of the octet (all lanes of the wave take part)
  float raw[NS];
  const int lane = threadIdx.x & 63, base_lane = lane & ~7;
#pragma unroll
  for (int i = 0; i < NS; i++) raw[i] = __shfl(cur, base_lane + i, 64);
  if (live) {
    // check kernel (entropy_gmm_table_cuda.cu:83-105): compares the raw entry with the already
    // shifted previous one, keeps every bin >= 1 count, takes the counts back from the widest bin
    float prev = 0.f, shift = 0.f, widest = 0.f, mine = 0.f;
    int widest_at = 0;
#pragma unroll
    for (int pt = 1; pt <= NS; pt++) {
      float c = raw[pt - 1];
      if (c <= prev) shift += 1;
      c += shift;
      if (c - prev > widest) {
        widest = c - prev;
        widest_at = pt - 1;
      }
      if (pt == q + 1) mine = c;
      prev = c;
    }
    if (shift > 0 && q >= widest_at) mine = (float)(int32_t)mine - shift;
    int32_t *row = table + ((size_t)n * len + l) * (NS + 1);
    row[q + 1] = (int32_t)mine;
    if (q == 0) row[0] = 0;
  }
  if (flags) {
    __syncthreads();  // every wave of the block has drained its stores (vmcnt(0) before the barrier)
    if (threadIdx.x == 0) {
      __threadfence_system();
      const int nblocks = gridDim.x * gridDim.y;
      if (atomicAdd(counter, 1) == nblocks - 1) {
        *counter = 0;  // ready for the next step of this stream
        __threadfence_system();
        __hip_atomic_store(const_cast<int32_t *>(flags) + 1, publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// all symbols at once, rows in stream order [step][img][position in the step's window]
__global__ void ee_tables_bulk_kernel(EeGeom g, const float *__restrict__ y, const float *__restrict__ symbols,
                                      int32_t *__restrict__ table, int32_t *__restrict__ labels, int nstep,
                                      float bias, float total, float beta, long long count) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    const int idx = (int)(i % g.npos);
    const int tc = (int)((i / g.npos) % g.ngroup);
    const int n = (int)(i / g.npos / g.ngroup);
    const int plane = g.pos_plane[idx];
    const int s = plane + tc;
    const int rows = g.h * g.npart;
    const int st = s - g.ngroup + 1 < 0 ? 0 : s - g.ngroup + 1;
    const int end = s < rows + g.w - 2 ? s + 1 : rows + g.w - 1;
    const int len = g.plane_start[end] - g.plane_start[st];
    const size_t r = (size_t)g.step_row[s] + (size_t)n * len + (idx - g.plane_start[st]);
    const Pos p = decode_pos(g.order[idx], g.h, g.w);
    const int cout = g.ngroup * 3;
    float par[3][3];
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {
      const float *base =
          y + ((((size_t)(rep * g.nimg + n) * g.npart + p.tg) * g.h + p.th) * g.w + p.tw) * cout + tc * 3;
#pragma unroll
      for (int k = 0; k < 3; k++) par[rep][k] = base[k];
    }
    gmm_prepare_row(par[0], par[1], 3, beta);
    gmm_cdf_row<int32_t>(par[0], par[1], par[2], 3, nstep, bias, total, 1, table + r * (nstep + 1));
    labels[r] = (int32_t)symbols[((((size_t)n * g.npart + p.tg) * g.ngroup + tc) * g.h + p.th) * g.w + p.tw];
  }
}

}  // namespace

int ee_tables_bulk(const EeGeom *g, const float *y_last, const float *symbols, int32_t *table, int32_t *labels,
                   int nstep, float bias, float total, float beta, void *stream) {
  const long long count = (long long)g->nimg * g->ngroup * g->npos;
  hipLaunchKernelGGL(ee_tables_bulk_kernel, dim3(pconv_grid(count)), dim3(256), 0, as_stream(stream), *g, y_last,
                     symbols, table, labels, nstep, bias, total, beta, count);
  PCONV_LAUNCH_CHECK("ee_tables_bulk");
  return PCONV_OK;
}

int ee_pack_weight(const float *w, float *packed, int nset, int cout, int cin, int ngroup, int constrain,
                   void *stream) {
  PCONV_REQUIRE(cout == GO * ngroup && cin % ngroup == 0 && (constrain == 5 || constrain == 6),
                "ee_pack_weight: bad layer shape");
  const int total = nset * (cout / GO) * slab_floats(cin);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), w, packed, cin,
                     ngroup, constrain == 5 ? 0 : 1, total);
  PCONV_LAUNCH_CHECK("ee_pack_weight");
  return PCONV_OK;
}

namespace {

// raises a kernel's dynamic-LDS limit once per device
template <class Kernel>
int band_lds_limit(Kernel kernel, size_t bytes, std::atomic<unsigned long long> &raised) {
  if (bytes <= 48 * 1024) return PCONV_OK;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) device = 0;
  const unsigned long long bit = 1ULL << (device & 63);
  if (raised.load(std::memory_order_acquire) & bit) return PCONV_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e != hipSuccess) {
    pconv_set_error("ee_conv: cannot raise dynamic LDS to %zu: %s", bytes, hipGetErrorString(e));
    return PCONV_ELAUNCH;
  }
  raised.fetch_or(bit, std::memory_order_release);
  return PCONV_OK;
}

template <int CIN, int NPL, int NB>
int band_step_launch(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
                     const float *slope, const float *residual, float *y, int pad_out, int slack, int first_plane,
                     int nplane, int psum, void *stream) {
  typedef Band<CIN, NPL, NB, false> B;
  static std::atomic<unsigned long long> raised{0};
  auto kernel = ee_band_step_kernel<CIN, NPL, NB>;
  static_assert(B::LDS_BYTES <= 160 * 1024, "the band fits the LDS of a CU");
  if (int rc = band_lds_limit(kernel, B::LDS_BYTES, raised)) return rc;
  const dim3 grid((unsigned)(g->npart * ((g->h + NB - 1) / NB)), (unsigned)((nplane + NPL - 1) / NPL),
                  (unsigned)(3 * g->nimg));
  PCONV_REQUIRE(grid.z <= 65535u && grid.y <= 65535u, "ee_conv: too many images / planes for one launch");
  hipLaunchKernelGGL(kernel, grid, dim3(B::BLOCK), B::LDS_BYTES, as_stream(stream), *g, x, shared_input, packed_w, bias,
                     slope, residual, y, pad_out, slack, first_plane, nplane, psum);
  return PCONV_OK;
}

template <int CIN, int NPL, int NB>
int band_bulk_launch(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
                     const float *slope, const float *residual, float *y, int pad_out, int slack, void *stream) {
  typedef Band<CIN, NPL, NB, true> B;
  static std::atomic<unsigned long long> raised{0};
  auto kernel = ee_band_bulk_kernel<CIN, NPL, NB>;
  static_assert(B::LDS_BYTES <= 160 * 1024, "band + slabs + output tile fit the LDS of a CU");
  if (int rc = band_lds_limit(kernel, B::LDS_BYTES, raised)) return rc;
  const int nplane = g->h * g->npart + g->w - 1;
  const dim3 grid((unsigned)(g->npart * ((g->h + NB - 1) / NB)), (unsigned)((nplane + NPL - 1) / NPL),
                  (unsigned)(3 * g->nimg));
  PCONV_REQUIRE(grid.z <= 65535u && grid.y <= 65535u, "ee_conv_bulk: too many images / planes for one launch");
  hipLaunchKernelGGL(kernel, grid, dim3(B::BLOCK), B::LDS_BYTES, as_stream(stream), *g, x, shared_input, packed_w, bias,
                     slope, residual, y, pad_out, slack, nplane);
  return PCONV_OK;
}

}  // namespace

int ee_conv(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
            const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
            int first_plane, int nplane, int psum, void *stream) {
  if (nplane <= 0) return PCONV_OK;
  PCONV_REQUIRE(cout == 3 * g->ngroup, "ee_conv: cout must be 3 per group");
  PCONV_REQUIRE(cin == g->ngroup || cin == 3 * g->ngroup, "ee_conv: cin must be 1 or 3 per group");
  const int slack = constrain == 5 ? 0 : 1;  // (the causal mask is also part of the packed slab's order)
  // PCONV_EE_ROWS: rows of a tile's diagonal per workgroup (8: more, shorter workgroups -- the step is a
  // latency chain; 16: one band serves twice the positions)
  static const int rows_env = getenv("PCONV_EE_ROWS") ? atoi(getenv("PCONV_EE_ROWS")) : 0;
  const int nb = rows_env == 16 || rows_env == 8 ? rows_env : 8;
#define EE_STEP(CIN)                                                                                                 \
  (nb == 16 ? band_step_launch<CIN, 4, 16>(g, x, shared_input, packed_w, bias, slope, residual, y, pad_out, slack,   \
                                           first_plane, nplane, psum, stream)                                        \
            : band_step_launch<CIN, 4, 8>(g, x, shared_input, packed_w, bias, slope, residual, y, pad_out, slack,    \
                                          first_plane, nplane, psum, stream))
  int rc;
  switch (cin) {
    case 14: rc = EE_STEP(14); break;
    case 42: rc = EE_STEP(42); break;
    case 28: rc = EE_STEP(28); break;
    case 84: rc = EE_STEP(84); break;
    case 48: rc = EE_STEP(48); break;
    case 144: rc = band_step_launch<144, 4, 8>(g, x, shared_input, packed_w, bias, slope, residual, y, pad_out, slack,
                                               first_plane, nplane, psum, stream); break;
    default:
      pconv_set_error("ee_conv: %d input channels not instantiated (14/42, 28/84, 48/144)", cin);
      return PCONV_EINVAL;
  }
#undef EE_STEP
  if (rc < 0) return rc;
  PCONV_LAUNCH_CHECK("ee_conv");
  return PCONV_OK;
}

int ee_conv_bulk(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
                 const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
                 void *stream) {
  PCONV_REQUIRE(cout == 3 * g->ngroup, "ee_conv_bulk: cout must be 3 per group");
  const int slack = constrain == 5 ? 0 : 1;
#define EE_BULK(CIN, NPL, NB) \
  band_bulk_launch<CIN, NPL, NB>(g, x, shared_input, packed_w, bias, slope, residual, y, pad_out, slack, stream)
  int rc;
  switch (cin) {
    case 14: rc = EE_BULK(14, 8, 8); break;
    case 42: rc = EE_BULK(42, 8, 8); break;
    case 28: rc = EE_BULK(28, 8, 8); break;
    case 48: rc = EE_BULK(48, 8, 8); break;
    case 84: rc = EE_BULK(84, 4, 8); break;
    case 144: rc = EE_BULK(144, 4, 8); break;
    default:
      pconv_set_error("ee_conv_bulk: %d input channels not instantiated (14/42, 28/84, 48/144)", cin);
      return PCONV_EINVAL;
  }
#undef EE_BULK
  if (rc < 0) return rc;
  PCONV_LAUNCH_CHECK("ee_conv_bulk");
  return PCONV_OK;
}

int ee_halo_bulk(const EeGeom *g, float *buf, int C, int nrep, void *stream) {
  const long long n_halo = (long long)nrep * g->npart * 2 * PAD * g->w * C;
  const long long n_wrap = (long long)nrep * g->npart * g->h * PAD * C;
  hipLaunchKernelGGL(ee_halo_bulk_kernel, dim3(pconv_grid(n_halo + n_wrap)), dim3(256), 0, as_stream(stream), *g,
                     buf, C, n_halo, n_wrap);
  PCONV_LAUNCH_CHECK("ee_halo_bulk");
  return PCONV_OK;
}

int ee_scatter(const EeGeom *g, const float *packed, float *ctx, int lo, int len, int psum, float bias,
               int32_t *flags, int32_t *relay, int wait_for, void *stream) {
  if (len <= 0) return PCONV_OK;
  hipLaunchKernelGGL(ee_scatter_kernel, dim3((len + 255) / 256, g->nimg), dim3(256), 0, as_stream(stream), *g, packed,
                     ctx, lo, len, psum, bias, (volatile int32_t *)flags, relay, wait_for);
  PCONV_LAUNCH_CHECK("ee_scatter");
  return PCONV_OK;
}

int ee_fill_ctx(const EeGeom *g, const float *symbols, float *ctx, float bias, void *stream) {
  const long long total = (long long)g->nimg * g->npart * g->ngroup * g->h * g->w;
  hipLaunchKernelGGL(ee_fill_ctx_kernel, dim3(pconv_grid(total)), dim3(256), 0, as_stream(stream), *g, symbols, ctx,
                     bias, total);
  PCONV_LAUNCH_CHECK("ee_fill_ctx");
  return PCONV_OK;
}

int ee_read_symbols(const EeGeom *g, const float *ctx, float *symbols, float bias, void *stream) {
  const long long total = (long long)g->nimg * g->npart * g->ngroup * g->h * g->w;
  hipLaunchKernelGGL(ee_read_symbols_kernel, dim3(pconv_grid(total)), dim3(256), 0, as_stream(stream), *g, ctx,
                     symbols, bias, total);
  PCONV_LAUNCH_CHECK("ee_read_symbols");
  return PCONV_OK;
}

int ee_tables(const EeGeom *g, const float *y_last, const float *symbols, int32_t *table, int32_t *labels, int lo,
              int len, int psum, int nstep, float bias, float total, float beta, int32_t *counter, int32_t *flags,
              int publish, void *stream) {
  if (len <= 0) return PCONV_OK;
  if (nstep == 8 && !symbols) {  // the decoder's form: 8 lanes per row
    hipLaunchKernelGGL(ee_tables8_kernel, dim3((len * 8 + 255) / 256, g->nimg), dim3(256), 0, as_stream(stream), *g,
                       y_last, table, lo, len, psum, bias, total, beta, counter, (volatile int32_t *)flags, publish);
    PCONV_LAUNCH_CHECK("ee_tables");
    return PCONV_OK;
  }
  hipLaunchKernelGGL(ee_tables_kernel, dim3((len + 255) / 256, g->nimg), dim3(256), 0, as_stream(stream), *g, y_last,
                     symbols, table, labels, lo, len, psum, nstep, bias, total, beta, counter,
                     (volatile int32_t *)flags, publish);
  PCONV_LAUNCH_CHECK("ee_tables");
  return PCONV_OK;
}
