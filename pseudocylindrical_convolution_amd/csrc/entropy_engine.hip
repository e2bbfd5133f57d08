// Channels-last kernels of the native entropy engine (see ee_kernels.h).
//
// Same arithmetic as the per-op kernels of entropy.hip (the streams must be
// byte-identical), different memory layout: with [tile][row][col][C] storage the
// 5 x 5 x C window of a position is 5 contiguous runs, and the reduction index
// kk = tap*C + ci walks memory in order, so a wave's 64 gathers hit 2-3 cache
// lines instead of ~13 with NCHW (the per-op layout), where this step kernel is
// bound by the number of cache lines the texture path has to look up.
//
// Halo taps are never stored: they are evaluated from the neighbouring tile's
// interior with the causal rule of pconv_host_causal_table (what
// EntropyCtxPadRun2 stores in the per-op path), which removes one launch per
// layer per step.
#include "common.h"
#include "ee_kernels.h"
#include "gmm_device.h"

namespace {

constexpr int kWave = 64;
#ifndef EE_CONV_BLOCK
#define EE_CONV_BLOCK 256
#endif
#ifndef EE_POS_PER_WAVE
#define EE_POS_PER_WAVE 2
#endif
#ifndef EE_WAVES_PER_EU
#define EE_WAVES_PER_EU 4
#endif
constexpr int kConvBlock = EE_CONV_BLOCK;  // one wavefront position per wave
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

// Packed weights: for every (set, output group) one slab [kk][4] holding the GO = 3
// rows of the group interleaved (4th float is padding), kk = tap*cin + ci.  A lane
// then fetches its three weights of a tap with one 16-byte LDS read.
__host__ __device__ constexpr int slab_floats(int cin) { return cin * KK * 4; }

__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ packed, int cin, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int red = cin * KK;
  const int o = i & 3, kk = (i >> 2) % red, grp = (i >> 2) / red;  // grp = set*ngroup + tc
  packed[i] = o < GO ? w[((size_t)grp * GO + o) * red + (kk % cin) * KK + kk / cin] : 0.f;
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

template <int CIN, int BLOCK>
__device__ __forceinline__ void stage_weights(float *wl, const float *__restrict__ wrow, int tid) {
  constexpr int N4 = slab_floats(CIN) / 4;
  const float4 *src = reinterpret_cast<const float4 *>(wrow);
  float4 *dst = reinterpret_cast<float4 *>(wl);
  for (int i = tid; i < N4; i += BLOCK) dst[i] = src[i];
}

// Step form: grid = 3 weight sets x planes of the step's window x `split`.  A
// workgroup stages the weight rows of its (set, plane) pair ONCE (all positions
// of a plane share the output group) and its waves then walk the plane's
// (image, position) list with stride split*waves.  Measured on MI355X: staging
// the 12.6 KB slab per 2 positions (one position per wave, small workgroups) made
// the L2->LDS copy 64 % of the kernel time; amortising it over a whole list
// slice removes that.
template <int CIN, int ITER, int BLOCK>
__global__ __launch_bounds__(BLOCK, EE_WAVES_PER_EU) void ee_conv_kernel(
    EeGeom g, const float *__restrict__ x, int shared_input, const float *__restrict__ wp,
    const float *__restrict__ bias, const float *__restrict__ slope, const float *__restrict__ residual,
    float *__restrict__ y, int cout, int constrain, int pad_out, int first_plane, int nplane, int split,
    int psum) {
  constexpr int RED = CIN * KK;
  constexpr int kWaves = BLOCK / kWave;
  __shared__ __attribute__((aligned(16))) float wl[slab_floats(CIN)];
  const int part = blockIdx.x % split;
  const int pl = (blockIdx.x / split) % nplane;
  const int set = blockIdx.x / split / nplane;
  const int plane = first_plane + pl;
  const int lo = g.plane_start[plane];
  const int cnt = g.plane_start[plane + 1] - lo;
  const int total = cnt * g.nimg;  // (image, position) pairs of this set and plane
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  if (part * kWaves >= total) return;  // uniform for the workgroup
  const int tc = psum - plane;
  const int group_in = CIN / g.ngroup;
#ifndef EE_ABL_NOSTAGE
  stage_weights<CIN, BLOCK>(wl, wp + ((size_t)set * g.ngroup + tc) * slab_floats(CIN), threadIdx.x);
#endif
  const int h = g.h, w = g.w;
  const int win = w + 2 * PAD;
  const int tile_elems = (h + 2 * PAD) * win * CIN;
  const int slack = (constrain == 5) ? 0 : 1;
  // causality: input group gi at (qh, pw) is usable iff gi + qh + pw < psum
  // (constrain 5) or <= psum (constrain 6); qh + pw = row + tw - 4 + kh + kw, so
  // the tap is usable iff ci < (tc + 4 - kh - kw + slack)*group_in.
  const int causal_base = (tc + slack) * group_in;
  __syncthreads();  // weight rows are in LDS; the waves run independently from here
#pragma unroll 1
  for (int e = part * kWaves + wave; e < total; e += split * kWaves) {
    const int img = e / cnt;
    const Pos p = decode_pos(g.order[lo + e - img * cnt], h, w);
    const int pn = set * g.nimg + img;  // replica-major image index
    const int xi = shared_input ? img : pn;
    const float *ximg = x + (size_t)xi * g.npart * tile_elems;
    const int valid = g.widths[p.tg];
    const bool edge = (p.th < HALF) || (p.th >= h - HALF) || (p.tw + HALF >= valid);
    float xv[ITER];  // masked taps hold 0: fmaf(0, w, acc) leaves acc unchanged
    // The tap walk depends on the lane only, but it is recomputed per position (a
    // few VALU ops per tap): letting the compiler keep it in ~100 loop-invariant
    // registers halves the occupancy and measured slower.
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    TapWalk<CIN> tw(lane_t);
    if (!edge) {
      // window origin (th-2+PAD, tw-2+PAD) = (th, tw) in padded coordinates
      const float *xin = ximg + (size_t)p.tg * tile_elems + ((size_t)p.th * win + p.tw) * CIN;
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const int kk = lane_t + it * kWave;
        const bool ok = (kk < RED) && (tw.lim(group_in) + causal_base > 0);
        xv[it] = ok ? xin[tw.off(win)] : 0.f;
        tw.next();
      }
    } else {
      const int rows = h * g.npart;
      int src_off[ITER], src_off1[ITER];  // element offsets inside the image, -1 = zero
      float src_w[ITER];
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const int kk = lane_t + it * kWave;
        const int kh = tw.kh, kw = tw.kw, ci = tw.ci;
        const bool ok = (kk < RED) && (tw.lim(group_in) + causal_base > 0);
        tw.next();
        src_off[it] = -1;
        src_off1[it] = -1;
        src_w[it] = 1.f;
        if (ok) {
          const int pr = p.th + kh;  // padded coordinates of the tap
          int pc = p.tw + kw;
          if (pc >= valid + PAD) pc -= valid;  // circular wrap of the first columns
          if (pr >= PAD && pr < h + PAD) {
            src_off[it] = p.tg * tile_elems + (pr * win + pc) * CIN + ci;  // left halo columns hold zeros
          } else if (pc >= PAD) {
            const int side = pr >= h + PAD;
            const int r = side ? pr - (h + PAD) : pr;
            const int row = side ? (p.tg + 1) * h + r : p.tg * h - PAD + r;
            if (row >= 0 && row < rows) {
              const int en = ((p.tg * 2 + side) * PAD + r) * w + pc - PAD;
              const int c = g.vh_col[en];
              if (c != -2) {
                const int st = row / h;
                const int rbase = st * tile_elems + ((row - st * h + PAD) * win + PAD) * CIN + ci;
                const int wst = g.widths[st];
                int c1 = c + 1;
                c1 = c1 >= wst ? c1 - wst : c1;
                src_w[it] = g.vh_wgt[en];
                src_off[it] = (c < 0) ? -1 : rbase + c * CIN;
                src_off1[it] = rbase + c1 * CIN;
              }
            }
          }
        }
      }
#pragma unroll
      for (int it = 0; it < ITER; it++) {
        const float a = (src_off[it] >= 0) ? ximg[src_off[it]] : 0.f;
        float v = a;
        if (src_off1[it] >= 0) v = a * src_w[it] + ximg[src_off1[it]] * (1 - src_w[it]);
        xv[it] = v;
      }
    }
    float acc[GO];
#pragma unroll
    for (int o = 0; o < GO; o++) acc[o] = 0.f;
    // the weight reads are loop-invariant; keep them as LDS reads (one 16-byte read
    // per tap) instead of 51 hoisted registers: occupancy matters more
    int lane_w = lane;
    asm volatile("" : "+v"(lane_w));
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane_w + it * kWave;
      const int kc = kk < RED ? kk : RED - 1;
      const float4 wv = *reinterpret_cast<const float4 *>(wl + 4 * kc);
      acc[0] = fmaf(xv[it], wv.x, acc[0]);
      acc[1] = fmaf(xv[it], wv.y, acc[1]);
      acc[2] = fmaf(xv[it], wv.z, acc[2]);
    }
#pragma unroll
    for (int o = 0; o < GO; o++) {
      float v = acc[o];
      for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
      acc[o] = v;
    }
    if (lane < GO) {
      float v = acc[0];
#pragma unroll
      for (int o = 1; o < GO; o++) v = (lane == o) ? acc[o] : v;
      const int pout = tc * GO + lane;
      const int bidx = set * cout + pout;
      v = v + bias[bidx];
      if (slope && v < 0) v = v * slope[bidx];
      const size_t oidx = ((((size_t)pn * g.npart + p.tg) * (h + 2 * pad_out) + p.th + pad_out) * (w + 2 * pad_out) +
                           p.tw + pad_out) * cout + pout;
      if (residual) v = v + residual[oidx];
      y[oidx] = v;
    }
  }
}

// Encoder ("bulk") form of the same layer: every symbol is known, so a position
// can be evaluated for ALL its channel groups at once.  One wave per position
// gathers the 5 x 5 x CIN window a single time (halo taps resolved once), then
// walks the groups: the workgroup stages the group's weight rows in LDS and every
// wave runs the masked fmaf chain + butterfly for that group.  Per output the
// operations and their order are exactly those of the step kernel above
// (psum = plane + group), so encoder and decoder tables agree bit for bit.
template <int CIN, int ITER, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ee_conv_bulk_kernel(
    EeGeom g, const float *__restrict__ x, int shared_input, const float *__restrict__ wp,
    const float *__restrict__ bias, const float *__restrict__ slope, const float *__restrict__ residual,
    float *__restrict__ y, int cout, int constrain, int pad_out) {
  constexpr int RED = CIN * KK;
  constexpr int kPosPerWg = BLOCK / kWave;
  __shared__ __attribute__((aligned(16))) float wl[slab_floats(CIN)];
  const int nchunk = (g.npos + kPosPerWg - 1) / kPosPerWg;
  const int chunk = blockIdx.x % nchunk;
  const int pn = blockIdx.x / nchunk;  // replica-major image index, 0 .. 3*nimg
  const int set = pn / g.nimg;
  const int group_in = CIN / g.ngroup;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int idx = chunk * kPosPerWg + wave;
  const bool active = idx < g.npos;  // wave-uniform
  const Pos p = decode_pos(g.order[active ? idx : 0], g.h, g.w);
  const int h = g.h, w = g.w;
  const int win = w + 2 * PAD;
  const int tile_elems = (h + 2 * PAD) * win * CIN;
  const int xi = shared_input ? pn % g.nimg : pn;
  const float *ximg = x + (size_t)xi * g.npart * tile_elems;
  const int slack = (constrain == 5) ? 0 : 1;
  const int valid = g.widths[p.tg];
  const bool edge = (p.th < HALF) || (p.th >= h - HALF) || (p.tw + HALF >= valid);
  float xv[ITER];
  int lim[ITER];  // causal limit of the tap; very negative for lanes past the reduction length
  TapWalk<CIN> tw(lane);
  if (!edge) {
    const float *xin = ximg + (size_t)p.tg * tile_elems + ((size_t)p.th * win + p.tw) * CIN;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane + it * kWave;
      const bool in = active && kk < RED;
      lim[it] = in ? tw.lim(group_in) : -(1 << 30);
      xv[it] = in ? xin[tw.off(win)] : 0.f;
      tw.next();
    }
  } else {
    const int rows = h * g.npart;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane + it * kWave;
      const int kh = tw.kh, kw = tw.kw, ci = tw.ci;
      const bool in = active && kk < RED;
      lim[it] = in ? tw.lim(group_in) : -(1 << 30);
      tw.next();
      float v = 0.f;
      if (in) {
        const int pr = p.th + kh;  // padded coordinates of the tap
        int pc = p.tw + kw;
        if (pc >= valid + PAD) pc -= valid;  // circular wrap of the first columns
        if (pr >= PAD && pr < h + PAD) {
          v = ximg[p.tg * tile_elems + (pr * win + pc) * CIN + ci];  // left halo columns hold zeros
        } else if (pc >= PAD) {
          const int side = pr >= h + PAD;
          const int r = side ? pr - (h + PAD) : pr;
          const int row = side ? (p.tg + 1) * h + r : p.tg * h - PAD + r;
          if (row >= 0 && row < rows) {
            const int e = ((p.tg * 2 + side) * PAD + r) * w + pc - PAD;
            const int c = g.vh_col[e];
            if (c != -2) {
              const int st = row / h;
              const int rbase = st * tile_elems + ((row - st * h + PAD) * win + PAD) * CIN + ci;
              const int wst = g.widths[st];
              int c1 = c + 1;
              c1 = c1 >= wst ? c1 - wst : c1;
              const float t = g.vh_wgt[e];
              const float a = (c < 0) ? 0.f : ximg[rbase + c * CIN];
              v = a * t + ximg[rbase + c1 * CIN] * (1 - t);
            }
          }
        }
      }
      xv[it] = v;
    }
  }
  const size_t obase = ((((size_t)pn * g.npart + p.tg) * (h + 2 * pad_out) + p.th + pad_out) * (w + 2 * pad_out) +
                        p.tw + pad_out) * cout;
  for (int tc = 0; tc < g.ngroup; tc++) {
    __syncthreads();  // previous group's LDS reads are done
    stage_weights<CIN, BLOCK>(wl, wp + ((size_t)set * g.ngroup + tc) * slab_floats(CIN), threadIdx.x);
    __syncthreads();
    const int causal_base = (tc + slack) * group_in;
    float acc[GO];
#pragma unroll
    for (int o = 0; o < GO; o++) acc[o] = 0.f;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane + it * kWave;
      const int kc = kk < RED ? kk : RED - 1;
      const float xm = (lim[it] + causal_base > 0) ? xv[it] : 0.f;  // fmaf(0, w, acc) == acc
      const float4 wv = *reinterpret_cast<const float4 *>(wl + 4 * kc);
      acc[0] = fmaf(xm, wv.x, acc[0]);
      acc[1] = fmaf(xm, wv.y, acc[1]);
      acc[2] = fmaf(xm, wv.z, acc[2]);
    }
#pragma unroll
    for (int o = 0; o < GO; o++) {
      float v = acc[o];
      for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
      acc[o] = v;
    }
    if (active && lane < GO) {
      float v = acc[0];
#pragma unroll
      for (int o = 1; o < GO; o++) v = (lane == o) ? acc[o] : v;
      const int pout = tc * GO + lane;
      const int bidx = set * cout + pout;
      v = v + bias[bidx];
      if (slope && v < 0) v = v * slope[bidx];
      if (residual) v = v + residual[obase + pout];
      y[obase + pout] = v;
    }
  }
}

__global__ void ee_scatter_kernel(EeGeom g, const float *__restrict__ packed, float *__restrict__ ctx, int lo,
                                  int len, int psum, float bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len * g.nimg) return;
  const int l = i % len, n = i / len;
  const Pos p = decode_pos(g.order[lo + l], g.h, g.w);
  const int tc = psum - p.tw - p.row;
  ctx[((((size_t)n * g.npart + p.tg) * (g.h + 2 * PAD) + p.th + PAD) * (g.w + 2 * PAD) + p.tw + PAD) * g.ngroup + tc] =
      packed[i] + bias;
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

__global__ void ee_tables_kernel(EeGeom g, const float *__restrict__ y, const float *__restrict__ symbols,
                                 int32_t *__restrict__ table, int32_t *__restrict__ labels, int lo, int len,
                                 int psum, int nstep, float bias, float total, float beta) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= len * g.nimg) return;
  const int l = r % len, n = r / len;
  const Pos p = decode_pos(g.order[lo + l], g.h, g.w);
  const int tc = psum - p.tw - p.row;
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
  gmm_cdf_row<int32_t>(par[0], par[1], par[2], 3, nstep, bias, total, 1, table + (size_t)r * (nstep + 1));
  if (symbols)
    labels[r] = (int32_t)symbols[((((size_t)n * g.npart + p.tg) * g.ngroup + tc) * g.h + p.th) * g.w + p.tw];
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

int ee_pack_weight(const float *w, float *packed, int nset, int cout, int cin, void *stream) {
  const int total = nset * (cout / GO) * slab_floats(cin);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), w, packed, cin,
                     total);
  PCONV_LAUNCH_CHECK("ee_pack_weight");
  return PCONV_OK;
}

int ee_conv(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
            const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
            int first_plane, int nplane, int longest_plane, int psum, void *stream) {
  if (nplane <= 0 || longest_plane <= 0) return PCONV_OK;
  PCONV_REQUIRE(cout == 3 * g->ngroup, "ee_conv: cout must be 3 per group");
  constexpr int kWaves = kConvBlock / kWave;
  // workgroups per (set, plane): enough to fill the chip, few enough that each
  // staged weight slab serves several positions per wave
  const int pairs = longest_plane * g->nimg;
  int split = (pairs + kWaves * EE_POS_PER_WAVE - 1) / (kWaves * EE_POS_PER_WAVE);
  if (split < 1) split = 1;
  const long long grid = (long long)3 * nplane * split;
#define EE_LAUNCH(CIN, ITER)                                                                                  \
  hipLaunchKernelGGL((ee_conv_kernel<CIN, ITER, kConvBlock>), dim3((unsigned)grid), dim3(kConvBlock), 0,      \
                     as_stream(stream), *g, x, shared_input, packed_w, bias, slope, residual, y, cout, constrain, \
                     pad_out, first_plane, nplane, split, psum)
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
  PCONV_LAUNCH_CHECK("ee_conv");
  return PCONV_OK;
}

int ee_conv_bulk(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
                 const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
                 void *stream) {
  PCONV_REQUIRE(cout == 3 * g->ngroup, "ee_conv_bulk: cout must be 3 per group");
  constexpr int kBlock = 1024;  // 16 positions share each staged weight slab
  const long long nchunk = (g->npos + kBlock / kWave - 1) / (kBlock / kWave);
  const long long grid = (long long)3 * g->nimg * nchunk;
  PCONV_REQUIRE(grid > 0 && grid < (1LL << 31), "ee_conv_bulk: grid %lld out of range", grid);
#define EE_BULK(CIN, ITER)                                                                                   \
  hipLaunchKernelGGL((ee_conv_bulk_kernel<CIN, ITER, kBlock>), dim3((unsigned)grid), dim3(kBlock), 0,        \
                     as_stream(stream), *g, x, shared_input, packed_w, bias, slope, residual, y, cout, constrain, \
                     pad_out)
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

int ee_scatter(const EeGeom *g, const float *packed, float *ctx, int lo, int len, int psum, float bias,
               void *stream) {
  if (len <= 0) return PCONV_OK;
  const int n = len * g->nimg;
  hipLaunchKernelGGL(ee_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), *g, packed, ctx, lo,
                     len, psum, bias);
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
              int len, int psum, int nstep, float bias, float total, float beta, void *stream) {
  if (len <= 0) return PCONV_OK;
  const int n = len * g->nimg;
  hipLaunchKernelGGL(ee_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), *g, y_last, symbols,
                     table, labels, lo, len, psum, nstep, bias, total, beta);
  PCONV_LAUNCH_CHECK("ee_tables");
  return PCONV_OK;
}
