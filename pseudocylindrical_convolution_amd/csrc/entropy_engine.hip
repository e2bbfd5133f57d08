// Channels-last kernels of the native entropy engine (see ee_kernels.h).
//
// Same arithmetic as the per-op kernels of entropy.hip (the streams must be
// byte-identical), different memory layout: with [tile][row][col][C] storage the
// 5 x 5 x C window of a position is 5 contiguous runs, and the reduction index
// kk = tap*C + ci walks memory in order, so a wave's 64 gathers hit 2-3 cache
// lines instead of ~13 with NCHW (the per-op layout), where this step kernel is
// bound by the number of cache lines the texture path has to look up.
//
// Halos are written by the producer of the interior value they derive from (see
// ee_kernels.h): the causal rule of pconv_host_causal_table (what
// EntropyCtxPadRun2 stores in the per-op path, one launch per layer per step) is
// applied in the epilogue of the kernel that computes the value, so consumers
// read plain padded windows.
#include <stdlib.h>
#include "common.h"
#include "ee_kernels.h"
#include "gmm_device.h"

namespace {

constexpr int kWave = 64;
constexpr int kConvBlock = 256;   // step kernel: 4 waves, one wavefront position per wave at a time
constexpr int kPosPerWave = 4;    // positions a wave walks with its weights in registers (measured best of 1..8)
// register cap of the step kernel: weights + offsets + window = 5 registers per tap
// and lane, so wider layers get fewer, fatter waves
constexpr int waves_per_eu(int iter) { return iter <= 20 ? 4 : (iter <= 40 ? 2 : 1); }
constexpr int K = 5, KK = 25, HALF = 2, PAD = 2, GO = 3;

struct Pos {
  int tw, row, tg, th;
};

// An integer CDF row of the codec's shape (8 symbols, total 65536) as the coder's packed 16-byte row
// (include/pconv_coder.h): uint16 c1 .. c7, then an auxiliary word -- bits 0-7 the label, bit 7 + k: c_k == 65536
// (stored as 0), bit 15: the row does not have this shape (the coder then refuses it, as it refuses an int32 row
// whose total is off).  16 bytes per symbol cross PCIe instead of 36 + 4.  A label outside [0, 255] (the int32 rows
// carry it unclipped and the coder answers "symbol out of range") is stored as 255, which the coder refuses the same way.
__device__ __forceinline__ uint4 pack_row16(const int32_t *row, int32_t label) {
  unsigned aux = (label < 0 || label > 255) ? 0xffu : (unsigned)label;
  unsigned hw[7];
  bool ok = row[0] == 0 && row[8] == 65536;
#pragma unroll
  for (int k = 1; k < 8; k++) {
    const unsigned v = (unsigned)row[k];
    ok = ok && v <= 65536u;
    hw[k - 1] = v & 0xffffu;
    aux |= (v == 65536u ? 1u : 0u) << (7 + k);
  }
  if (!ok) aux |= 0x8000u;
  return make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (aux << 16));
}
__device__ __forceinline__ Pos decode_pos(int hw, int h, int w) {
  Pos p;
  p.tw = hw % w;
  p.row = hw / w;
  p.tg = p.row / h;
  p.th = p.row - p.tg * h;
  return p;
}

// v[l] + v[l ^ off] for off = 32, 16, 8, 4, 2, 1 -- the canonical butterfly of the
// masked convolution (every lane ends with the same total, bit for bit what
// __shfl_xor gives: each step adds the same two numbers) -- on the cross-lane
// VALU paths of gfx950 instead of six ds_bpermute round trips: half / row swaps,
// a row rotate, one ds_swizzle (xor 4 has no DPP form) and two quad permutes.
__device__ __forceinline__ float butterfly_sum(float v) {
  {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F));                      // xor 4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad [1,0,3,2]
  return v;
}

// The three outputs of one position in one packed butterfly: 7 exchange-adds instead
// of 18, same pairs and order per value.  Result per lane: the total of output
// {0, 2, 1, 2}[lane >> 4] (rows of 16 lanes).
__device__ __forceinline__ float butterfly3(float v0, float v1, float v2) {
  float a01, a2;
  {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
    a01 = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // lanes 0-31: v0, lanes 32-63: v1
    const unsigned u = __float_as_uint(v2);
    auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    a2 = __uint_as_float(q[0]) + __uint_as_float(q[1]);   // v2 in both halves
  }
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a01), __float_as_uint(a2), false, false);
  float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // rows: v0, v2, v1, v2
  t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));  // xor 8
  t += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(t), 0x101F));                      // xor 4
  t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x4E, 0xF, 0xF, false));   // xor 2
  t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xF, 0xF, false));   // xor 1
  return t;
}

// The same butterfly for 12 values at once (4 positions x 3 outputs of the bulk kernel).
// A plain butterfly repeats every exchange in both partners; here a step keeps each
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

// Packed weights: for every (set, output group) one slab [slot][4] holding the GO = 3
// rows of the group interleaved (4th float is padding), slot = tap*cin + ci, padded to
// whole waves, with the causal mask of the output group already applied (see
// ee_kernels.h).  A lane fetches its three weights of a tap with one 16-byte LDS read and
// a masked tap contributes fmaf(x, 0, acc) == acc (x is finite).
__host__ __device__ constexpr int slab_slots(int cin) { return (cin * KK + kWave - 1) / kWave * kWave; }
__host__ __device__ constexpr int slab_floats(int cin) { return slab_slots(cin) * 4; }

__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ packed, int cin, int ngroup,
                                   int slack, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int red = cin * KK, slots = (red + kWave - 1) / kWave * kWave;
  const int o = i & 3, kk = (i >> 2) % slots, grp = (i >> 2) / slots;  // grp = set*ngroup + tc
  const int tc = grp % ngroup, group_in = cin / ngroup;
  const int tap = kk / cin, ci = kk - tap * cin;
  const int kh = tap / K, kw = tap - kh * K;
  // causality: input group gi at window offset (kh, kw) is usable iff gi + kh + kw - 4 < tc + slack
  const bool ok = kk < red && o < GO && (2 * HALF - kh - kw) * group_in - ci + (tc + slack) * group_in > 0;
  packed[i] = ok ? w[((size_t)grp * GO + o) * red + ci * KK + tap] : 0.f;
}

// Walks the reduction index kk = lane, lane + 64, ... and keeps its decomposition
// kk = (kh*5 + kw)*CIN + ci up to date with a few adds (no division, no table):
// one wave-level VALU op is much cheaper here than one more vector memory
// instruction -- the step kernels are bound by the number of those.
template <int CIN>
struct TapWalk {
  int ci, kh, kw;
  __device__ __forceinline__ explicit TapWalk(int lane) {
    ci = lane % CIN;
    const int tap = lane / CIN;
    kw = tap % K;
    kh = tap / K;
  }
  __device__ __forceinline__ void next() {
    constexpr int Q = kWave / CIN, R = kWave % CIN;
    ci += R;
    int inc = Q;
    if (ci >= CIN) {
      ci -= CIN;
      inc++;
    }
    kw += inc;
#pragma unroll
    for (int k = 0; k < (Q + 1 + K - 1) / K; k++)
      if (kw >= K) {
        kw -= K;
        kh++;
      }
  }
  // element offset of the tap from the window origin / causal limit
  __device__ __forceinline__ int off(int win) const { return (kh * win + kw) * CIN + ci; }
  __device__ __forceinline__ int lim(int group_in) const { return (2 * HALF - kh - kw) * group_in - ci; }
};

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

// copies a (pre-masked, padded) slab to LDS
template <int CIN, int BLOCK>
__device__ __forceinline__ void stage_weights(float *wl, const float *__restrict__ wrow, int tid) {
  constexpr int N4 = slab_floats(CIN) / 4;
  const float4 *src = reinterpret_cast<const float4 *>(wrow);
  float4 *dst = reinterpret_cast<float4 *>(wl);
  for (int i = tid; i < N4; i += BLOCK) dst[i] = src[i];
}

// Step form: grid = (parts, planes of the step's window, 3 weight sets x images).  All
// positions of a plane share the output group, so the workgroup stages the (set, group)
// slab ONCE, by LDS-DMA (no staging registers, no VALU); a lane keeps the byte offsets
// of its ITER taps inside a window in registers (a table: they depend on the layer type
// only) and the waves then walk the plane's position list with stride parts*waves.  Per
// position: one 16-byte scalar load of its record (the next one is requested before the
// current one is used), ITER gathers from one scalar base, 3*ITER fmaf against the LDS
// slab, the packed butterfly, the epilogue.  The launch is a chain of memory round trips
// per wave, so what matters is how many waves are resident (<= 64 registers: 8 per SIMD)
// and how few dependent hops a position takes: no integer division anywhere (3-D grid,
// precomputed records), halo entries written from 16-byte records (2 hops instead of 4).
template <int CIN, int ITER, int BLOCK, int PJ>
__global__ __launch_bounds__(BLOCK, BLOCK > 512 ? 4 : (ITER * PJ <= 20 ? 8 : (ITER * PJ <= 40 ? (PJ > 1 ? 6 : 4) : 2))) void ee_step_kernel(
    EeGeom g, const float *__restrict__ x, int shared_input, const float *__restrict__ wp,
    const uint32_t *__restrict__ tapoff, const float *__restrict__ bias, const float *__restrict__ slope,
    const float *__restrict__ residual, float *__restrict__ y, int pad_out, int first_plane, int psum,
    int contiguous, int xcd_split, int xcd_nplane) {
  constexpr int kWaves = BLOCK / kWave;
  constexpr int SLOTS = ITER * kWave;
  static_assert(SLOTS == slab_slots(CIN), "ITER must cover the padded reduction length");
  // Which (part, plane, set, image) a workgroup is.  3-D grid: read off blockIdx.  xcd_split > 0 (r6): a 1-D grid
  // in XCD-MAJOR order -- workgroup b runs on XCD b % 8 (round-robin dispatch; an assumption for speed only), and the
  // workgroups of one XCD take a CONTIGUOUS range of the (set, plane, image, part) order, so that the ~42 / 8 weight
  // slabs a launch's XCD needs are fetched into ITS L2 once instead of all 42 into every L2 (every launch starts
  // with cold L2s: the slabs were 7/8 of the re-fetched bytes), and neighbouring planes of one set -- whose
  // windows overlap -- meet in one L2.
  int part, split, plane, pn;
  if (xcd_split > 0) {
    const int npn = 3 * g.nimg;
    const int nb = xcd_split * xcd_nplane * npn;
    const int per = (nb + 7) >> 3;
    const int v = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (v >= nb) return;  // (uniform)
    split = xcd_split;
    part = v % split;
    int q = v / split;
    const int im = q % g.nimg;
    q /= g.nimg;
    plane = first_plane + q % xcd_nplane;
    pn = (q / xcd_nplane) * g.nimg + im;
  } else {
    part = blockIdx.x, split = gridDim.x;
    plane = first_plane + blockIdx.y;
    pn = blockIdx.z;  // replica-major image index: set * nimg + img
  }
  const int set = (pn >= g.nimg) + (pn >= 2 * g.nimg);
  const int img = pn - set * g.nimg;
  typedef const __attribute__((address_space(4))) int32_t const_i32_t;
  const_i32_t *pstart = (const_i32_t *)g.plane_start;
  const int lo = pstart[plane];
  const int cnt = pstart[plane + 1] - lo;
  // A workgroup takes a CONTIGUOUS share of the plane's position list (neighbours on the
  // anti-diagonal: their 5 x 5 windows overlap, 16 of 25 taps between direct neighbours, so what
  // one wave gathered the next finds in the CU's L1) -- the kernel is bound by the bytes its
  // gathers pull out of L2.  contiguous = 0: the interleaved assignment (position e of a wave,
  // e + waves*parts the next one).
  const int share = contiguous ? (cnt + split - 1) / split : cnt;
  const int first = contiguous ? part * share : part * kWaves;
  const int last = contiguous ? (first + share < cnt ? first + share : cnt) : cnt;
  if (first >= cnt) return;  // uniform for the workgroup
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int tc = psum - plane;
  const int cout = GO * g.ngroup;
  const int h = g.h, w = g.w;
  const int win = w + 2 * PAD;
  __shared__ __attribute__((aligned(16))) float4 lw[SLOTS];
  {
    typedef __attribute__((address_space(3))) void lds_ptr_t;
    typedef const __attribute__((address_space(1))) void glb_ptr_t;
    const float4 *slab = reinterpret_cast<const float4 *>(wp + ((size_t)set * g.ngroup + tc) * slab_floats(CIN));
    for (int i = wave; i < ITER; i += kWaves)
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)(slab + i * kWave + lane), (lds_ptr_t *)(lw + i * kWave), 16, 0, 0);
  }
  // byte offsets of a lane's ITER taps inside a window: unsigned 32-bit, so the gathers are "scalar
  // base + lane offset" loads.  One position at a time they live in registers; with PJ > 1 the
  // registers go to the second window and the table sits in LDS (read once per PJ positions)
  unsigned off[PJ > 1 ? 1 : ITER];
  __shared__ unsigned toff[PJ > 1 ? SLOTS : 1];
  if (PJ > 1) {
    typedef __attribute__((address_space(3))) void lds_ptr_t;
    typedef const __attribute__((address_space(1))) void glb_ptr_t;
    for (int i = wave; i < ITER; i += kWaves)
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)(tapoff + i * kWave + lane), (lds_ptr_t *)(toff + i * kWave), 4, 0, 0);
  } else {
#pragma unroll
    for (int it = 0; it < ITER; it++) off[it] = tapoff[it * kWave + lane];
  }
  const int pout0 = tc * GO;
  const int bidx = set * cout + pout0;
  const float b0 = bias[bidx], b1 = bias[bidx + 1], b2 = bias[bidx + 2];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (slope) {
    s0 = slope[bidx];
    s1 = slope[bidx + 1];
    s2 = slope[bidx + 2];
  }
  const size_t in_img = (size_t)g.npart * (h + 2 * PAD) * win * CIN;
  const size_t out_img = (size_t)g.npart * (h + 2 * pad_out) * (w + 2 * pad_out) * cout;
  const float *ximg = x + (size_t)(shared_input ? img : pn) * in_img;
  float *yimg = y + (size_t)pn * out_img;
  const float *rimg = residual ? residual + (size_t)pn * out_img : nullptr;
  __syncthreads();  // (waits for the DMA: vmcnt(0))
  // contiguous == 2 (r6): the PJ positions a wave takes through the loop body together are NEIGHBOURS on the
  // anti-diagonal (e, e + 1: 16 of their 25 taps are the same bytes, requested back to back by one wave, so the second
  // window's lines are in the CU's L1 or in flight); contiguous == 1: e and e + waves (round 3-5: one shared tap).
  // Which wave evaluates a position changes nothing in its arithmetic.
  const int stride = contiguous ? kWaves : split * kWaves;
  const int js = contiguous == 2 ? 1 : stride;
  int e = first + (contiguous == 2 ? wave * PJ : wave);
  if (e >= last) return;
  // read-only table, wave-uniform index: through the constant address space these are
  // scalar loads (s_load_dwordx4), not a vector load + readfirstlane
  const_i32_t *plist = (const_i32_t *)(g.pos + lo);
  auto load_pos = [&](int i) {
    EePos p;
    p.pix = plist[4 * i];
    p.hw = plist[4 * i + 1];
    p.wrap = plist[4 * i + 2];
    p.rev = plist[4 * i + 3];
    return p;
  };
  // PJ positions (e, e + stride, ...) go through the loop body together: their gathers are in
  // flight at the same time, ONE pass over the LDS slab feeds all their fmaf chains (the slab reads
  // were as many LDS cycles as the fmafs were VALU cycles), and a wave's chain of dependent round
  // trips is PJ times shorter.  Per position the operations and their order are unchanged.
  EePos rec[PJ];
#pragma unroll
  for (int j = 0; j < PJ; j++) rec[j] = load_pos(e + j * js < last ? e + j * js : e);
#pragma unroll 1
  for (;;) {
    const int en = e + PJ * stride;
    EePos nxt[PJ];  // requested now, used by the next iteration
#pragma unroll
    for (int j = 0; j < PJ; j++) nxt[j] = load_pos(en + j * js < last ? en + j * js : e);
    float xv[PJ][ITER];
    size_t oflat[PJ];
    float r0[PJ], r1[PJ], r2[PJ];  // issued with the gathers: one memory round trip per position
    const float *xin[PJ];
#pragma unroll
    for (int j = 0; j < PJ; j++) {
      xin[j] = ximg + (size_t)rec[j].pix * CIN;  // window origin (row-2, col-2) in padded coordinates
      oflat[j] = (size_t)(pad_out ? rec[j].pix + 2 * win + 2 : rec[j].hw) * cout + pout0;
    }
    int tl = lane;  // (opaque: the table reads below must not be hoisted out of the position loop into registers)
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      // (opaque to the optimiser: a zero-extension hoisted out of the loop would
      // turn every gather into a 64-bit VALU add + a 2-register address)
      unsigned o = PJ > 1 ? toff[tl + it * kWave] : off[PJ > 1 ? 0 : it];
      asm volatile("" : "+v"(o));
#pragma unroll
      for (int j = 0; j < PJ; j++)
        xv[j][it] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xin[j]) + o);
    }
#pragma unroll
    for (int j = 0; j < PJ; j++) {
      r0[j] = r1[j] = r2[j] = 0.f;
      if (rimg) {
        r0[j] = rimg[oflat[j]];
        r1[j] = rimg[oflat[j] + 1];
        r2[j] = rimg[oflat[j] + 2];
      }
    }
    float a0[PJ], a1[PJ], a2[PJ];
    if constexpr (PJ == 2) {
      // the two positions' chains as packed fp32 FMAs (v_pk_fma_f32: two IEEE fmas per instruction,
      // the same bits as two v_fma_f32)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const float4 wv = lw[lane + it * kWave];
        asm volatile("" ::"v"(wv.w));  // (see below)
        const f2 xx = {xv[0][it], xv[1][it]};
        p0 = __builtin_elementwise_fma(xx, (f2){wv.x, wv.x}, p0);
        p1 = __builtin_elementwise_fma(xx, (f2){wv.y, wv.y}, p1);
        p2 = __builtin_elementwise_fma(xx, (f2){wv.z, wv.z}, p2);
        if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      a0[0] = p0.x, a0[1] = p0.y, a1[0] = p1.x, a1[1] = p1.y, a2[0] = p2.x, a2[1] = p2.y;
    } else {
#pragma unroll
      for (int j = 0; j < PJ; j++) a0[j] = a1[j] = a2[j] = 0.f;
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const float4 wv = lw[lane + it * kWave];
        // (the padding float kept live: a 16-byte ds_read_b128 takes 4 LDS cycles, the
        // 12-byte ds_read_b96 the compiler would otherwise pick takes 8)
        asm volatile("" ::"v"(wv.w));
#pragma unroll
        for (int j = 0; j < PJ; j++) {
          a0[j] = fmaf(xv[j][it], wv.x, a0[j]);
          a1[j] = fmaf(xv[j][it], wv.y, a1[j]);
          a2[j] = fmaf(xv[j][it], wv.z, a2[j]);
        }
      }
    }
    // one packed butterfly per position; the row of 16 lanes a lane sits in decides which output
    // it finishes (rows 0 / 2 / 1,3 -> outputs 0 / 1 / 2), and the epilogue is spread the same way
    const int row = lane >> 4;
    const int o = row == 0 ? 0 : (row == 2 ? 1 : 2);
    const bool writer = (lane & 15) == 0 && row != 3;  // one lane per output
    float vsum[PJ];
#pragma unroll
    for (int j = 0; j < PJ; j++) {
      vsum[j] = butterfly3(a0[j], a1[j], a2[j]);
      // (opaque: otherwise the whole chain of a position that may not exist -- gathers, slab reads,
      // fmafs -- is sunk into the branch that stores it, and the positions run one after the other)
      asm volatile("" : "+v"(vsum[j]));
    }
#pragma unroll
    for (int j = 0; j < PJ; j++) {
      if (j > 0 && e + j * js >= last) break;  // (uniform) no such position: its lanes computed a copy of e
      float v = vsum[j] + (o == 0 ? b0 : (o == 1 ? b1 : b2));
      if (slope) {
        const float sl = o == 0 ? s0 : (o == 1 ? s1 : s2);
        v = v < 0 ? v * sl : v;
      }
      if (rimg) v = v + (o == 0 ? r0[j] : (o == 1 ? r1[j] : r2[j]));
      if (writer) yimg[oflat[j] + o] = v;
      if (pad_out) {
        if (rec[j].wrap && writer) yimg[oflat[j] + (size_t)rec[j].wrap * cout + o] = v;  // circular wrap copy
        const int nrev = rec[j].rev & 15;
        if (nrev && row != 3) {
          // this value feeds halo rows of the neighbouring tiles: the 16 lanes of a row share
          // the entries interpolated from it
          const EeHalo *hr = g.halo + (rec[j].rev >> 4);
          const int ch = pout0 + o;
          for (int k = lane & 15; k < nrev; k += 16) {
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
    if (en >= last) break;
#pragma unroll
    for (int j = 0; j < PJ; j++) rec[j] = nxt[j];
    e = en;
  }
}

// Decoder, LAST layer of a step and the CDF rows of its positions in ONE launch (r6): the last
// layer has no activation, no residual and no padded output -- its 3 x 3 values per (position, group) are read by
// the table kernel only.  A workgroup is (part, plane, image) and owns ALL THREE weight sets of its positions (three
// slabs in LDS): a wave takes two neighbouring positions through the three sets one after the other -- per (position,
// set) the very gathers, packed fmaf chains and butterfly of ee_step_kernel<CIN, ITER, BLOCK, 2> -- then the nine
// parameters of a position sit in the wave (a row of 16 lanes per output, identical in its lanes: three v_readlane per
// set), lanes 0-7 / 8-15 evaluate the 8 CDF entries of the first / second position with ee_tables8_kernel's
// operations (same device functions, same repair), and the octet's first lane stores the packed 16-byte row.  One
// launch boundary and one latency-bound kernel less per step; identical rows by construction.
template <int CIN, int ITER, int BLOCK>
__global__ __launch_bounds__(BLOCK, 2) void ee_step_tables_kernel(
    EeGeom g, const float *__restrict__ x, const float *__restrict__ wp, const uint32_t *__restrict__ tapoff,
    const float *__restrict__ bias, int32_t *__restrict__ table, int first_plane, int psum, int win_lo, int win_len,
    float gbias, float total, float beta, int32_t *counter, volatile int32_t *flags, int publish) {
  constexpr int kWaves = BLOCK / kWave;
  constexpr int SLOTS = ITER * kWave;
  constexpr int NS = 8;
  static_assert(SLOTS == slab_slots(CIN), "ITER must cover the padded reduction length");
  const int part = blockIdx.x, split = gridDim.x;
  const int plane = first_plane + blockIdx.y;
  const int img = blockIdx.z;
  typedef const __attribute__((address_space(4))) int32_t const_i32_t;
  const_i32_t *pstart = (const_i32_t *)g.plane_start;
  const int lo = pstart[plane];
  const int cnt = pstart[plane + 1] - lo;
  const int share = (cnt + split - 1) / split;
  const int first = part * share;
  const int last = first + share < cnt ? first + share : cnt;
  const bool active = first < cnt;  // (uniform; an idle workgroup still counts itself in at the end)
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int tc = psum - plane;
  const int cout = GO * g.ngroup;
  const int h = g.h, w = g.w;
  const int win = w + 2 * PAD;
  __shared__ __attribute__((aligned(16))) float4 lw[3][SLOTS];
  __shared__ unsigned toff[SLOTS];
  if (active) {
    typedef __attribute__((address_space(3))) void lds_ptr_t;
    typedef const __attribute__((address_space(1))) void glb_ptr_t;
#pragma unroll
    for (int set = 0; set < 3; set++) {
      const float4 *slab = reinterpret_cast<const float4 *>(wp + ((size_t)set * g.ngroup + tc) * slab_floats(CIN));
      for (int i = wave; i < ITER; i += kWaves)
        __builtin_amdgcn_global_load_lds((glb_ptr_t *)(slab + i * kWave + lane), (lds_ptr_t *)(&lw[set][i * kWave]), 16, 0, 0);
    }
    for (int i = wave; i < ITER; i += kWaves)
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)(tapoff + i * kWave + lane), (lds_ptr_t *)(toff + i * kWave), 4, 0, 0);
  }
  const int row = lane >> 4;
  const int o = row == 0 ? 0 : (row == 2 ? 1 : 2);
  float bo[3];
#pragma unroll
  for (int set = 0; set < 3; set++) bo[set] = bias[set * cout + tc * GO + o];
  const size_t in_img = (size_t)g.npart * (h + 2 * PAD) * win * CIN;
  __syncthreads();  // (waits for the DMA: vmcnt(0))
  if (active) {
    const_i32_t *plist = (const_i32_t *)(g.pos + lo);
    for (int e = first + wave * 2; e < last; e += 2 * kWaves) {
      const bool second = e + 1 < last;  // (uniform)
      const int e1 = second ? e + 1 : e;
      const int pix0 = plist[4 * e], pix1 = plist[4 * e1];
      float val[3][2];
#pragma unroll
      for (int set = 0; set < 3; set++) {
        const float *ximg = x + (size_t)(set * g.nimg + img) * in_img;
        const float *xin0 = ximg + (size_t)pix0 * CIN, *xin1 = ximg + (size_t)pix1 * CIN;
        float xv[2][ITER];
        int tl = lane;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int it = 0; it < ITER; it++) {
          unsigned off = toff[tl + it * kWave];
          asm volatile("" : "+v"(off));
          xv[0][it] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xin0) + off);
          xv[1][it] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xin1) + off);
        }
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
        for (int it = 0; it < ITER; it++) {
          const float4 wv = lw[set][lane + it * kWave];
          asm volatile("" ::"v"(wv.w));
          const f2 xx = {xv[0][it], xv[1][it]};
          p0 = __builtin_elementwise_fma(xx, (f2){wv.x, wv.x}, p0);
          p1 = __builtin_elementwise_fma(xx, (f2){wv.y, wv.y}, p1);
          p2 = __builtin_elementwise_fma(xx, (f2){wv.z, wv.z}, p2);
          if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        float s0 = butterfly3(p0.x, p1.x, p2.x), s1 = butterfly3(p0.y, p1.y, p2.y);
        asm volatile("" : "+v"(s0), "+v"(s1));
        val[set][0] = s0 + bo[set];
        val[set][1] = s1 + bo[set];
      }
      // the nine parameters of each position: output k of a set sits in the lanes of row {0, 2, 1}[k]
      const int pj = (lane >> 3) & 1;
      float par[3][3];
#pragma unroll
      for (int set = 0; set < 3; set++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int src = k == 0 ? 0 : (k == 1 ? 32 : 16);
          const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val[set][0]), src));
          const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val[set][1]), src));
          par[set][k] = pj ? b : a;
        }
      }
      const int q = lane & 7;
      const bool have = lane < 16 && (pj == 0 || second);
      gmm_prepare_row(par[0], par[1], 3, beta);
      float cur = gmm_cdf_entry(par[0], par[1], par[2], 3, NS, q + 1, gbias, total);
      float raw[NS];
      const int base_lane = lane & ~7;
#pragma unroll
      for (int i = 0; i < NS; i++) raw[i] = __shfl(cur, base_lane + i, 64);
      {
        // check kernel (entropy_gmm_table_cuda.cu:83-105), as ee_tables8_kernel
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
        cur = mine;
      }
      int32_t rowv[NS + 1];
      rowv[0] = 0;
#pragma unroll
      for (int i = 0; i < NS; i++) rowv[i + 1] = (int32_t)__shfl(cur, base_lane + i, 64);
      if (have && q == 0) {
        const int l = lo + e + pj - win_lo;  // row of this position in the step's window
        reinterpret_cast<uint4 *>(table)[(size_t)img * win_len + l] = pack_row16(rowv, 0);
      }
    }
  }
  if (flags) {
    __syncthreads();  // every wave of the block has drained its stores (vmcnt(0) before the barrier)
    if (threadIdx.x == 0) {
      __threadfence_system();
      const int nblocks = gridDim.x * gridDim.y * gridDim.z;
      if (atomicAdd(counter, 1) == nblocks - 1) {
        *counter = 0;  // ready for the next step of this stream
        __threadfence_system();
        __hip_atomic_store(const_cast<int32_t *>(flags) + 1, publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// Encoder ("bulk") form of the same layer: every symbol is known, so a position
// can be evaluated for ALL its channel groups at once.  A wave owns PP positions:
// it gathers their 5 x 5 x CIN windows a single time, then walks the groups -- the
// workgroup stages the group's (pre-masked) slab in LDS, every lane reads its taps'
// weights once (causally masked taps are zeros) and feeds the PP masked fmaf chains
// + butterflies of its positions.  Per output the operations and their order are
// exactly those of the step kernel above (psum = plane + group), so encoder and
// decoder tables agree bit for bit.  Halos of the output are filled afterwards by
// ee_halo_bulk.
template <int CIN, int ITER, int BLOCK, int PP>
__global__ __launch_bounds__(BLOCK) void ee_conv_bulk_kernel(
    EeGeom g, const float *__restrict__ x, int shared_input, const float *__restrict__ wp,
    const float *__restrict__ bias, const float *__restrict__ slope, const float *__restrict__ residual,
    float *__restrict__ y, int cout, int constrain, int pad_out, int first_idx, int n_idx, int s_lo, int s_hi) {
  // Step range [s_lo, s_hi) (the whole schedule: 0, INT_MAX): only the (position, group) pairs whose wavefront
  // step plane + group lies in it are evaluated and stored; first_idx / n_idx = the schedule entries that have
  // such a pair (planes s_lo - ngroup + 1 .. s_hi - 1).  The encoder's last group is coded in step ranges so
  // that the arithmetic coder starts on the first range while the GPU is on the second (engine.cpp).
  constexpr int RED = CIN * KK;
  constexpr int kPosPerWg = BLOCK / kWave * PP;
  __shared__ __attribute__((aligned(16))) float wl2[2][slab_floats(CIN)];  // double-buffered weight slab
  const int nchunk = (n_idx + kPosPerWg - 1) / kPosPerWg;
  const int chunk = blockIdx.x % nchunk;
  const int pn = blockIdx.x / nchunk;  // replica-major image index, 0 .. 3*nimg
  const int end_idx = first_idx + n_idx;
  const int set = pn / g.nimg;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int h = g.h, w = g.w;
  const int win = w + 2 * PAD;
  const int tile_elems = (h + 2 * PAD) * win * CIN;
  const int xi = shared_input ? pn % g.nimg : pn;
  const float *ximg = x + (size_t)xi * g.npart * tile_elems;
  const int idx0 = first_idx + (chunk * (BLOCK / kWave) + wave) * PP;
  // groups any position of this workgroup has in the range (the schedule is sorted by plane): uniform loop bounds
  const int wg_first = first_idx + chunk * kPosPerWg;
  const int wg_last = wg_first + kPosPerWg - 1 < end_idx - 1 ? wg_first + kPosPerWg - 1 : end_idx - 1;
  const int plane_min = g.pos_plane[wg_first], plane_max = g.pos_plane[wg_last];
  const int tc_lo = s_lo - plane_max > 0 ? s_lo - plane_max : 0;
  const int tc_hi = s_hi - plane_min < g.ngroup ? s_hi - plane_min : g.ngroup;
  if (tc_lo >= tc_hi) return;  // (uniform for the workgroup)
  unsigned off[ITER];  // byte offsets of this lane's taps inside a window
  {
    TapWalk<CIN> tw(lane);
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      off[it] = (lane + it * kWave < RED) ? 4u * (unsigned)tw.off(win) : 0u;
      tw.next();
    }
  }
  float xv[PP][ITER];
  size_t obase[PP];
  int pplane[PP];
#pragma unroll
  for (int j = 0; j < PP; j++) {
    const int idx = idx0 + j < end_idx ? idx0 + j : first_idx;
    pplane[j] = __builtin_amdgcn_readfirstlane(g.pos_plane[idx]);
    const Pos p = decode_pos(__builtin_amdgcn_readfirstlane(g.order[idx]), h, w);
    const float *xin = ximg + (size_t)p.tg * tile_elems + ((size_t)p.th * win + p.tw) * CIN;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      unsigned o = off[it];
      asm volatile("" : "+v"(o));  // keeps the zero-extension out of a hoisted 64-bit add (see ee_conv_kernel)
      const float v = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xin) + o);
      // lanes past the reduction length (last iteration only) contribute fmaf(0, w, acc) == acc
      xv[j][it] = ((it + 1) * kWave <= RED || lane + it * kWave < RED) ? v : 0.f;
    }
    obase[j] = ((((size_t)pn * g.npart + p.tg) * (h + 2 * pad_out) + p.th + pad_out) * (w + 2 * pad_out) +
                p.tw + pad_out) * cout;
  }
  stage_weights<CIN, BLOCK>(wl2[tc_lo & 1], wp + ((size_t)set * g.ngroup + tc_lo) * slab_floats(CIN), threadIdx.x);
  // bias and slope of the set's outputs in LDS, once: loaded per group in front of their use they sat behind the
  // store of the group before (vmcnt counts stores too, and a guarded load is waited for with vmcnt(0)): one
  // exposed store -> load round trip per group and wave
  __shared__ float bs_tab[2][GO * CIN];  // (cout = 3 ngroup, ngroup = CIN or CIN / 3)
  for (int i = threadIdx.x; i < cout; i += BLOCK) {
    bs_tab[0][i] = bias[set * cout + i];
    bs_tab[1][i] = slope ? slope[set * cout + i] : 1.f;  // (v * 1 is v)
  }
  __syncthreads();
  for (int tc = tc_lo; tc < tc_hi; tc++) {
    // the next group's slab goes to the other buffer while this one is used (its
    // last readers passed the barrier that ended the previous iteration)
    if (tc + 1 < tc_hi)
      stage_weights<CIN, BLOCK>(wl2[(tc + 1) & 1], wp + ((size_t)set * g.ngroup + tc + 1) * slab_floats(CIN),
                                threadIdx.x);
    const float *wl = wl2[tc & 1];
    float acc[PP][GO];
    if constexpr (PP == 4) {
      // the four positions' chains as packed fp32 FMAs, two positions per instruction (v_pk_fma_f32: two
      // IEEE fmas, the same bits as two v_fma_f32) -- r3: the loop was 12 scalar FMAs per LDS read, VALU-bound
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p[2][GO];
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int o = 0; o < GO; o++) p[q][o] = (f2){0.f, 0.f};
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const int kk = lane + it * kWave;
        const int kc = kk < RED ? kk : RED - 1;
        const float4 wv = *reinterpret_cast<const float4 *>(wl + 4 * kc);  // causally masked taps are zeros
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const f2 xx = {xv[2 * q][it], xv[2 * q + 1][it]};
          p[q][0] = __builtin_elementwise_fma(xx, (f2){wv.x, wv.x}, p[q][0]);
          p[q][1] = __builtin_elementwise_fma(xx, (f2){wv.y, wv.y}, p[q][1]);
          p[q][2] = __builtin_elementwise_fma(xx, (f2){wv.z, wv.z}, p[q][2]);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int o = 0; o < GO; o++) {
          acc[2 * q][o] = p[q][o].x;
          acc[2 * q + 1][o] = p[q][o].y;
        }
    } else {
#pragma unroll
      for (int j = 0; j < PP; j++)
#pragma unroll
        for (int o = 0; o < GO; o++) acc[j][o] = 0.f;
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const int kk = lane + it * kWave;
        const int kc = kk < RED ? kk : RED - 1;
        const float4 wv = *reinterpret_cast<const float4 *>(wl + 4 * kc);  // causally masked taps are zeros
#pragma unroll
        for (int j = 0; j < PP; j++) {
          acc[j][0] = fmaf(xv[j][it], wv.x, acc[j][0]);
          acc[j][1] = fmaf(xv[j][it], wv.y, acc[j][1]);
          acc[j][2] = fmaf(xv[j][it], wv.z, acc[j][2]);
        }
      }
    }
    if constexpr (PP == 4) {
      // all 12 reductions at once; lane L ends with position L >> 4, output {0,2,1,2}[quad]
      const float tot = butterfly12(acc, lane);
      const int j = lane >> 4, quad = (lane >> 2) & 3;
      const int o = quad == 0 ? 0 : (quad == 2 ? 1 : 2);
      const int pl = j == 0 ? pplane[0] : (j == 1 ? pplane[1] : (j == 2 ? pplane[2] : pplane[3]));
      if ((lane & 3) == 0 && quad != 3 && idx0 + j < end_idx && pl + tc >= s_lo && pl + tc < s_hi) {
        const int pout = tc * GO + o;
        const size_t ob = j == 0 ? obase[0] : (j == 1 ? obase[1] : (j == 2 ? obase[2] : obase[3]));
        float v = tot + bs_tab[0][pout];
        if (v < 0) v = v * bs_tab[1][pout];
        if (residual) v = v + residual[ob + pout];
        y[ob + pout] = v;
      }
    } else {
      const int pout = tc * GO + (lane < GO ? lane : 0);
      const int bidx = set * cout + pout;
      const float bv = bias[bidx];
      const float sv = slope ? slope[bidx] : 0.f;
#pragma unroll
      for (int j = 0; j < PP; j++) {
#pragma unroll
        for (int o = 0; o < GO; o++) acc[j][o] = butterfly_sum(acc[j][o]);
        if (idx0 + j < end_idx && lane < GO && pplane[j] + tc >= s_lo && pplane[j] + tc < s_hi) {
          float v = acc[j][0];
#pragma unroll
          for (int o = 1; o < GO; o++) v = (lane == o) ? acc[j][o] : v;
          v = v + bv;
          if (slope && v < 0) v = v * sv;
          if (residual) v = v + residual[obase[j] + pout];
          y[obase[j] + pout] = v;
        }
      }
    }
    __syncthreads();
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
                                  volatile int32_t *flags, int publish, int packed) {
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
    if (!packed) {
      int32_t *row = table + ((size_t)n * len + l) * (NS + 1);
      row[q + 1] = (int32_t)mine;
      if (q == 0) row[0] = 0;
    }
    cur = mine;
  }
  if (packed) {
    // the octet's eight entries meet in every lane; its first lane stores the row as ONE 16-byte piece (the rows go
    // straight to pinned host memory: one PCIe write per row instead of nine 4-byte ones)
    int32_t row[NS + 1];
    row[0] = 0;
#pragma unroll
    for (int i = 0; i < NS; i++) row[i + 1] = (int32_t)__shfl(cur, base_lane + i, 64);
    if (live && q == 0) reinterpret_cast<uint4 *>(table)[(size_t)n * len + l] = pack_row16(row, 0);
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
                                      float bias, float total, float beta, long long count, int first_idx, int n_idx,
                                      int s_lo, int s_hi, int packed) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    const int idx = first_idx + (int)(i % n_idx);
    const int tc = (int)((i / n_idx) % g.ngroup);
    const int n = (int)(i / n_idx / g.ngroup);
    const int plane = g.pos_plane[idx];
    const int s = plane + tc;
    if (s < s_lo || s >= s_hi) continue;  // (another step range's row)
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
    const int32_t label = (int32_t)symbols[((((size_t)n * g.npart + p.tg) * g.ngroup + tc) * g.h + p.th) * g.w + p.tw];
    if (packed) {  // (nstep == 8: checked on the host)
      int32_t row[9];
      gmm_cdf_row<int32_t>(par[0], par[1], par[2], 3, 8, bias, total, 1, row);
      reinterpret_cast<uint4 *>(table)[r] = pack_row16(row, label);
    } else {
      gmm_cdf_row<int32_t>(par[0], par[1], par[2], 3, nstep, bias, total, 1, table + r * (nstep + 1));
      labels[r] = label;
    }
  }
}

}  // namespace

int ee_tables_bulk(const EeGeom *g, const float *y_last, const float *symbols, int32_t *table, int32_t *labels,
                   int nstep, float bias, float total, float beta, int first_idx, int n_idx, int s_lo, int s_hi,
                   int packed, void *stream) {
  PCONV_REQUIRE(first_idx >= 0 && n_idx >= 0 && first_idx + n_idx <= g->npos && s_lo < s_hi, "ee_tables_bulk: bad range");
  PCONV_REQUIRE(!packed || (nstep == 8 && total == 65536.f), "ee_tables_bulk: packed rows are 8 symbols of total 65536");
  if (n_idx == 0) return PCONV_OK;
  const long long count = (long long)g->nimg * g->ngroup * n_idx;
  hipLaunchKernelGGL(ee_tables_bulk_kernel, dim3(pconv_grid(count)), dim3(256), 0, as_stream(stream), *g, y_last,
                     symbols, table, labels, nstep, bias, total, beta, count, first_idx, n_idx, s_lo, s_hi, packed);
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

int ee_conv(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
            const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
            int first_plane, int nplane, int longest_plane, int psum, void *stream) {
  if (nplane <= 0 || longest_plane <= 0) return PCONV_OK;
  PCONV_REQUIRE(cout == 3 * g->ngroup, "ee_conv: cout must be 3 per group");
  PCONV_REQUIRE(cin == g->ngroup || cin == 3 * g->ngroup, "ee_conv: cin must be 1 or 3 per group");
  (void)constrain;  // the causal mask is part of the packed slab
  // workgroups per (set, plane, image): enough to fill the chip, few enough that a staged
  // slab serves several positions.  PCONV_EE_BLOCK (threads per workgroup: 256 / 512 / 1024)
  // and PCONV_EE_PPW (positions a wave walks) are tuning knobs.
  static const int block = getenv("PCONV_EE_BLOCK") ? atoi(getenv("PCONV_EE_BLOCK")) : kConvBlock;
  // measured (MI355X, 4096x2048, decode of 1 / 2 / 4 / 8 frames in two groups, contiguous shares, two
  // positions per loop body): 2 positions per wave 93 / 109 / - / - ms, 4: 96 / 110 / 150 / 222,
  // 8: 116 / 126 / 140 / 211, 16: - / 137 / 164 / 213; 512- and 1024-thread workgroups are slower
  // at every batch size
  static const int ppw_env = getenv("PCONV_EE_PPW") ? atoi(getenv("PCONV_EE_PPW")) : 0;
  const int ppw = ppw_env > 0 ? ppw_env : (g->nimg <= 1 ? 2 : 2 * kPosPerWave);
  // PCONV_EE_JOINT: positions a wave takes through the loop body together (1 or 2)
  static const int joint = getenv("PCONV_EE_JOINT") ? atoi(getenv("PCONV_EE_JOINT")) : 2;
  // PCONV_EE_CONTIG: contiguous (1) or interleaved (0) shares of a plane per workgroup
  // (2 = contiguous shares AND the positions of one loop body neighbours on the anti-diagonal: r6)
  static const int contig = getenv("PCONV_EE_CONTIG") ? atoi(getenv("PCONV_EE_CONTIG")) : 1;
  const int waves = block / kWave;
  int split = (longest_plane + waves * ppw - 1) / (waves * ppw);
  if (split < 1) split = 1;
  const uint32_t *tap = cin == g->ngroup ? g->tap_in : g->tap_hid;
  // PCONV_EE_XCD: 1 = the 1-D XCD-major workgroup order (see the kernel), 0 (default) = the 3-D grid of rounds 3-5.
  // Measured (r6, profiles/round6_step_kernel_pmc.txt): XCD-major + neighbour pairs cut the L1 -> L2 requests by a
  // third and the fabric reads by a sixth, and the decode takes the same time at 8 frames (176-183 vs 176-180 ms) and
  // LONGER at one frame (95-97 vs 91 ms): the launch is a chain of dependent round trips, not a bandwidth problem.
  static const int xcd = getenv("PCONV_EE_XCD") ? atoi(getenv("PCONV_EE_XCD")) : 0;
  const long long nb = (long long)split * nplane * 3 * g->nimg;
  PCONV_REQUIRE(3 * g->nimg <= 65535 && nplane <= 65535 && nb < (1LL << 30), "ee_conv: too many images for one launch");
  const dim3 grid = xcd ? dim3((unsigned)(8 * ((nb + 7) / 8)), 1, 1) : dim3((unsigned)split, (unsigned)nplane, (unsigned)(3 * g->nimg));
  const int xs = xcd ? split : 0;
#define EE_LAUNCH_J(CIN, ITER, BLK, PJ)                                                                      \
  hipLaunchKernelGGL((ee_step_kernel<CIN, ITER, BLK, PJ>), grid, dim3(BLK), 0, as_stream(stream), *g, x,     \
                     shared_input, packed_w, tap, bias, slope, residual, y, pad_out, first_plane, psum, contig, xs, nplane)
#define EE_LAUNCH_B(CIN, ITER, BLK)                          \
  if (joint == 2 && ITER <= 20 && ppw >= 2) {                \
    EE_LAUNCH_J(CIN, ITER, BLK, (ITER <= 20 ? 2 : 1));       \
  } else {                                                   \
    EE_LAUNCH_J(CIN, ITER, BLK, 1);                          \
  }
#define EE_LAUNCH(CIN, ITER)              \
  if (block == 1024) {                    \
    EE_LAUNCH_B(CIN, ITER, 1024);         \
  } else if (block == 512) {              \
    EE_LAUNCH_B(CIN, ITER, 512);          \
  } else {                                \
    EE_LAUNCH_B(CIN, ITER, 256);          \
  }
  if (cin == 14) {
    EE_LAUNCH(14, 6);
  } else if (cin == 42) {
    EE_LAUNCH(42, 17);
  } else if (cin == 28) {
    EE_LAUNCH(28, 11);
  } else if (cin == 84) {
    EE_LAUNCH(84, 33);
  } else if (cin == 48) {
    EE_LAUNCH(48, 19);
  } else if (cin == 144) {
    EE_LAUNCH(144, 57);
  } else {
    pconv_set_error("ee_conv: %d input channels not instantiated (14/42, 28/84, 48/144)", cin);
    return PCONV_EINVAL;
  }
#undef EE_LAUNCH
#undef EE_LAUNCH_B
#undef EE_LAUNCH_J
  PCONV_LAUNCH_CHECK("ee_conv");
  return PCONV_OK;
}

int ee_conv_tables(const EeGeom *g, const float *x, const float *packed_w, const float *bias, int32_t *table, int cin,
                   int first_plane, int nplane, int longest_plane, int psum, int lo, int len, float gbias, float total,
                   float beta, int32_t *counter, int32_t *flags, int publish, void *stream) {
  if (nplane <= 0 || longest_plane <= 0 || len <= 0) return PCONV_OK;
  PCONV_REQUIRE(cin == 42 && g->ngroup == 14 && total == 65536.f, "ee_conv_tables: the codec's shape only (14 groups, 8 x 65536 rows)");
  // positions a wave takes: two per loop body; PCONV_EE_FUSE_PPW positions per wave (default 4: two bodies)
  static const int ppw = getenv("PCONV_EE_FUSE_PPW") ? atoi(getenv("PCONV_EE_FUSE_PPW")) : 4;
  const int waves = kConvBlock / kWave;
  int split = (longest_plane + waves * ppw - 1) / (waves * ppw);
  if (split < 1) split = 1;
  PCONV_REQUIRE(g->nimg <= 65535 && nplane <= 65535, "ee_conv_tables: too many images for one launch");
  const dim3 grid((unsigned)split, (unsigned)nplane, (unsigned)g->nimg);
  hipLaunchKernelGGL((ee_step_tables_kernel<42, 17, kConvBlock>), grid, dim3(kConvBlock), 0, as_stream(stream), *g, x, packed_w,
                     g->tap_hid, bias, table, first_plane, psum, lo, len, gbias, total, beta, counter,
                     (volatile int32_t *)flags, publish);
  PCONV_LAUNCH_CHECK("ee_conv_tables");
  return PCONV_OK;
}

int ee_conv_bulk(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
                 const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
                 int first_idx, int n_idx, int s_lo, int s_hi, void *stream) {
  PCONV_REQUIRE(cout == 3 * g->ngroup, "ee_conv_bulk: cout must be 3 per group");
  PCONV_REQUIRE(first_idx >= 0 && n_idx >= 0 && first_idx + n_idx <= g->npos && s_lo < s_hi, "ee_conv_bulk: bad range");
  if (n_idx == 0) return PCONV_OK;
  constexpr int kBlock = 1024;
  // positions per wave (each staged weight slab then serves 16x as many): as many as
  // fit 128 registers beside the window (ITER values per position)
  const int iter = (cin * KK + kWave - 1) / kWave;
  const int pp = iter <= 20 ? 4 : 1;
  const long long per_wg = kBlock / kWave * pp;
  const long long nchunk = (n_idx + per_wg - 1) / per_wg;
  const long long grid = (long long)3 * g->nimg * nchunk;
  PCONV_REQUIRE(grid > 0 && grid < (1LL << 31), "ee_conv_bulk: grid %lld out of range", grid);
#define EE_BULK(CIN, ITER)                                                                                   \
  hipLaunchKernelGGL((ee_conv_bulk_kernel<CIN, ITER, kBlock, (ITER <= 20 ? 4 : 1)>), dim3((unsigned)grid), dim3(kBlock), 0,        \
                     as_stream(stream), *g, x, shared_input, packed_w, bias, slope, residual, y, cout, constrain, \
                     pad_out, first_idx, n_idx, s_lo, s_hi)
  if (cin == 14) {
    EE_BULK(14, 6);
  } else if (cin == 42) {
    EE_BULK(42, 17);
  } else if (cin == 28) {
    EE_BULK(28, 11);
  } else if (cin == 84) {
    EE_BULK(84, 33);
  } else if (cin == 48) {
    EE_BULK(48, 19);
  } else if (cin == 144) {
    EE_BULK(144, 57);
  } else {
    pconv_set_error("ee_conv_bulk: %d input channels not instantiated (14/42, 28/84, 48/144)", cin);
    return PCONV_EINVAL;
  }
#undef EE_BULK
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
              int publish, int packed, void *stream) {
  if (len <= 0) return PCONV_OK;
  PCONV_REQUIRE(!packed || (nstep == 8 && !symbols && total == 65536.f), "ee_tables: packed rows are the decoder's 8 x 65536 rows");
  if (nstep == 8 && !symbols) {  // the decoder's form: 8 lanes per row
    hipLaunchKernelGGL(ee_tables8_kernel, dim3((len * 8 + 255) / 256, g->nimg), dim3(256), 0, as_stream(stream), *g,
                       y_last, table, lo, len, psum, bias, total, beta, counter, (volatile int32_t *)flags, publish, packed);
    PCONV_LAUNCH_CHECK("ee_tables");
    return PCONV_OK;
  }
  hipLaunchKernelGGL(ee_tables_kernel, dim3((len + 255) / 256, g->nimg), dim3(256), 0, as_stream(stream), *g, y_last,
                     symbols, table, labels, lo, len, psum, nstep, bias, total, beta, counter,
                     (volatile int32_t *)flags, publish);
  PCONV_LAUNCH_CHECK("ee_tables");
  return PCONV_OK;
}
