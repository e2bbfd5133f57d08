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
#define EE_CONV_BLOCK 128  /* measured on MI355X: 128 > 256 > 512 (finer packing, shorter barrier waits) */
#endif
#ifndef EE_WAVES_PER_EU
#define EE_WAVES_PER_EU 1
#endif
constexpr int kConvBlock = EE_CONV_BLOCK;  // one wavefront position per wave
constexpr int kBulkBlock = kEeBulkPos * kWave;
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

__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ packed, int cin, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int red = cin * KK;
  const int kk = i % red, row = i / red;
  packed[i] = w[(size_t)row * red + (kk % cin) * KK + kk / cin];
}

// BLOCK threads = BLOCK/64 positions per workgroup.  BULK: blockIdx enumerates
// (workgroup of the bulk map, group, image) and psum = plane + group.
template <int CIN, int ITER, int BLOCK, bool BULK>
__global__ __launch_bounds__(BLOCK, EE_WAVES_PER_EU) void ee_conv_kernel(
    EeGeom g, const float *__restrict__ x, int shared_input, const float *__restrict__ wp,
    const float *__restrict__ bias, const float *__restrict__ slope, const float *__restrict__ residual,
    float *__restrict__ y, int cout, int constrain, int pad_out, int first_plane, int nplane, int chunks,
    int psum) {
  constexpr int RED = CIN * KK;
  constexpr int kPosPerWg = BLOCK / kWave;
  __shared__ float wl[GO * RED];
  int plane, first, pn, tc;
  if (BULK) {
    const int wg = blockIdx.x % g.nbulk_wg;
    tc = (blockIdx.x / g.nbulk_wg) % g.ngroup;
    pn = blockIdx.x / g.nbulk_wg / g.ngroup;
    plane = g.bulk_wg[2 * wg];
    first = g.bulk_wg[2 * wg + 1];
    psum = plane + tc;
  } else {
    const int chunk = blockIdx.x % chunks;
    const int pl = (blockIdx.x / chunks) % nplane;
    pn = blockIdx.x / chunks / nplane;  // replica-major image index, 0 .. 3*nimg
    plane = first_plane + pl;
    first = chunk * kPosPerWg;
    tc = psum - plane;
  }
  const int lo = g.plane_start[plane];
  const int cnt = g.plane_start[plane + 1] - lo;
  if (first >= cnt) return;  // uniform for the workgroup
  const int set = pn / g.nimg;
  const int group_in = CIN / g.ngroup;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int pi = first + wave;
  const bool active = pi < cnt;  // wave-uniform
  const int hw = g.order[lo + (active ? pi : first)];
  {
    const float *wrow = wp + ((size_t)set * cout + tc * GO) * RED;
    for (int i = threadIdx.x; i < GO * RED; i += BLOCK) wl[i] = wrow[i];
  }
  const Pos p = decode_pos(hw, g.h, g.w);
  const int h = g.h, w = g.w;
  const int win = w + 2 * PAD;
  const int tile_elems = (h + 2 * PAD) * win * CIN;
  const int xi = shared_input ? pn % g.nimg : pn;
  const float *ximg = x + (size_t)xi * g.npart * tile_elems;
  const int slack = (constrain == 5) ? 0 : 1;
  const int valid = g.widths[p.tg];
  const bool edge = (p.th < HALF) || (p.th >= h - HALF) || (p.tw + HALF >= valid);
  // causality: input group gi at (qh, pw) is usable iff gi + qh + pw < psum
  // (constrain 5) or <= psum (constrain 6); qh + pw = row + tw - 4 + kh + kw, so
  // the tap is usable iff ci < (tc + 4 - kh - kw + slack)*group_in, i.e.
  // tap_lim[kk] + (tc + slack)*group_in > 0.  The index math of a tap depends on
  // the lane only and comes from precomputed tables.
  const int tt = (CIN == g.ngroup) ? 0 : 1;
  const int32_t *__restrict__ t_off = g.tap_off[tt];
  const int32_t *__restrict__ t_lim = g.tap_lim[tt];
  const int causal_base = (tc + slack) * group_in;
  float xv[ITER];
  bool ok[ITER];
  if (!edge) {
    // window origin (th-2+PAD, tw-2+PAD) = (th, tw) in padded coordinates
    const float *xin = ximg + (size_t)p.tg * tile_elems + ((size_t)p.th * win + p.tw) * CIN;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane + it * kWave;
      const int kc = kk < RED ? kk : RED - 1;
      ok[it] = active && (kk < RED) && (t_lim[kc] + causal_base > 0);
      xv[it] = ok[it] ? xin[t_off[kc]] : 0.f;
    }
  } else {
    const int rows = h * g.npart;
    int src_off[ITER], src_off1[ITER];  // element offsets inside the image, -1 = zero
    float src_w[ITER];
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane + it * kWave;
      const int kc = kk < RED ? kk : RED - 1;
      const int tp = g.tap_pos[tt][kc];
      const int kh = tp & 15, kw = (tp >> 4) & 15, ci = tp >> 8;
      ok[it] = active && (kk < RED) && (t_lim[kc] + causal_base > 0);
      src_off[it] = -1;
      src_off1[it] = -1;
      src_w[it] = 1.f;
      if (ok[it]) {
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
            const int e = ((p.tg * 2 + side) * PAD + r) * w + pc - PAD;
            const int c = g.vh_col[e];
            if (c != -2) {
              const int st = row / h;
              const int rbase = st * tile_elems + ((row - st * h + PAD) * win + PAD) * CIN + ci;
              const int wst = g.widths[st];
              int c1 = c + 1;
              c1 = c1 >= wst ? c1 - wst : c1;
              src_w[it] = g.vh_wgt[e];
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
  __syncthreads();  // weight rows are in LDS
  if (!active) return;
  float acc[GO];
#pragma unroll
  for (int o = 0; o < GO; o++) acc[o] = 0.f;
#pragma unroll
  for (int it = 0; it < ITER; it++) {
    const int kk = lane + it * kWave;
    const int kc = kk < RED ? kk : RED - 1;
#pragma unroll
    for (int o = 0; o < GO; o++) {
      const float f = fmaf(xv[it], wl[o * RED + kc], acc[o]);
      acc[o] = ok[it] ? f : acc[o];
    }
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
  const int total = nset * cout * cin * KK;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), w, packed, cin,
                     total);
  PCONV_LAUNCH_CHECK("ee_pack_weight");
  return PCONV_OK;
}

template <int BLOCK, bool BULK>
static int ee_conv_launch(const EeGeom *g, const float *x, int shared_input, const float *packed_w,
                          const float *bias, const float *slope, const float *residual, float *y, int cin, int cout,
                          int constrain, int pad_out, int first_plane, int nplane, int chunks, int psum,
                          long long grid, void *stream) {
  PCONV_REQUIRE(cout == 3 * g->ngroup, "ee_conv: cout must be 3 per group");
  PCONV_REQUIRE(grid > 0 && grid < (1LL << 31), "ee_conv: grid %lld out of range", grid);
#define EE_LAUNCH(CIN, ITER)                                                                                   \
  hipLaunchKernelGGL((ee_conv_kernel<CIN, ITER, BLOCK, BULK>), dim3((unsigned)grid), dim3(BLOCK), 0,            \
                     as_stream(stream), *g, x, shared_input, packed_w, bias, slope, residual, y, cout, constrain, \
                     pad_out, first_plane, nplane, chunks, psum)
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

int ee_conv(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
            const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
            int first_plane, int nplane, int longest_plane, int psum, void *stream) {
  if (nplane <= 0 || longest_plane <= 0) return PCONV_OK;
  constexpr int per_wg = kConvBlock / kWave;
  const int chunks = (longest_plane + per_wg - 1) / per_wg;
  return ee_conv_launch<kConvBlock, false>(g, x, shared_input, packed_w, bias, slope, residual, y, cin, cout,
                                           constrain, pad_out, first_plane, nplane, chunks, psum,
                                           (long long)3 * g->nimg * nplane * chunks, stream);
}

int ee_conv_bulk(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
                 const float *slope, const float *residual, float *y, int cin, int cout, int constrain, int pad_out,
                 void *stream) {
  return ee_conv_launch<kBulkBlock, true>(g, x, shared_input, packed_w, bias, slope, residual, y, cin, cout,
                                          constrain, pad_out, 0, 0, 0, 0,
                                          (long long)3 * g->nimg * g->ngroup * g->nbulk_wg, stream);
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
