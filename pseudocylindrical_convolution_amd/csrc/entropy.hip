// Entropy-model wavefront ops (one call = one step of one op), wave64 design.
//
// The context model walks the latent in 3-D anti-diagonal order: at step `psum`
// the symbols with group + row + col == psum are coded (row counts over the
// npart stacked tiles).  For 2-D plane p = row + col the channel group handled at
// this step is psum - p.  `order` lists positions sorted by plane
// (pconv_host_wavefront); a step touches the contiguous slice [lo, lo+len).
//
// reference: d_input_cuda_v2.cu, entropy_ctx_pad_run2_cuda.cu,
// entropy_conv_cuda_v2.cu, entropy_add_cuda.cu, d_extract_cuda_v2.cu,
// entropy_gmm_table_cuda.cu.
#include "common.h"
#include "../../include/pconv_detmath.h"
#include "gmm_device.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;

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

// d_input_cuda_v2.cu:32-52
__global__ __launch_bounds__(kBlock) void dinput2_kernel(const float *__restrict__ packed,
                                                         float *__restrict__ ctx,
                                                         const int32_t *__restrict__ order, int lo,
                                                         int len, int nimg, int ngroup, int npart,
                                                         int h, int w, int pad, int psum, float bias,
                                                         int rep) {
  const int total = len * nimg;
  const int hout = h + 2 * pad, wout = w + 2 * pad;
  const size_t rep_stride = (size_t)nimg * npart * ngroup * hout * wout;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const int tl = i % len, tn = i / len;
    const Pos p = decode_pos(order[lo + tl], h, w);
    const int tc = psum - p.tw - p.row;
    const size_t idx =
        ((((size_t)tn * npart + p.tg) * ngroup + tc) * hout + p.th + pad) * wout + p.tw + pad;
    const float v = packed[i] + bias;
    for (int j = 0; j < rep; j++) ctx[idx + j * rep_stride] = v;
  }
}

// entropy_ctx_pad_run2_cuda.cu:33-65 on the integer halo lists
__global__ __launch_bounds__(kBlock) void ctx_pad_run2_kernel(
    float *__restrict__ data, const int32_t *__restrict__ dst, const int32_t *__restrict__ src0,
    const int32_t *__restrict__ src1, const float *__restrict__ wgt,
    const int32_t *__restrict__ entry_plane, int lo, int len, int nimg, int cpn, size_t plane_sz,
    size_t img_stride, int psum) {
  const long long total = (long long)nimg * cpn * len;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int e = lo + (int)(i % len);
    const int ppc = (int)((i / len) % cpn);
    const long long tn = i / len / cpn;
    const int pc = (psum - entry_plane[e]) * cpn + ppc;
    const size_t base = (size_t)tn * img_stride + (size_t)pc * plane_sz;
    const int s0 = src0[e], s1 = src1[e];
    float v;
    if (s1 == -2) {
      v = data[base + s0];
    } else {
      const float t = wgt[e];
      const float a = (s0 < 0) ? 0.f : data[base + s0];
      v = a * t + data[base + s1] * (1 - t);
    }
    data[base + dst[e]] = v;
  }
}

// decoded symbols out of the context tensor: interior + bias, zero in the dead
// columns (pseudo_codec.py:159-160: b[:npart,:,2:-2,2:-2] + bias, then PseudoFill)
__global__ __launch_bounds__(kBlock) void ctx_to_symbols_kernel(const float *__restrict__ ctx,
                                                                float *__restrict__ out,
                                                                const int32_t *__restrict__ widths, int c,
                                                                int h, int w, int pad, int npart, float bias,
                                                                long long total) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int tw = (int)(i % w);
    const int th = (int)((i / w) % h);
    const long long plane = i / w / h;
    const int tg = (int)((plane / c) % npart);
    float v = 0.f;
    if (tw < widths[tg]) v = ctx[(plane * (h + 2 * pad) + th + pad) * (w + 2 * pad) + tw + pad] + bias;
    out[i] = v;
  }
}

// encoder side: all symbols are known, so the whole context tensor is filled once
// (value + bias inside the valid width, `rep` replicas); the causal mask of the
// convolutions keeps every step from seeing more than DInput2 would have given it
__global__ __launch_bounds__(kBlock) void symbols_to_ctx_kernel(const float *__restrict__ sym,
                                                                float *__restrict__ ctx,
                                                                const int32_t *__restrict__ widths, int c,
                                                                int h, int w, int pad, int npart, float bias,
                                                                int rep, long long total) {
  const size_t rep_stride = (size_t)total / ((size_t)h * w) * (h + 2 * pad) * (w + 2 * pad);
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int tw = (int)(i % w);
    const int th = (int)((i / w) % h);
    const long long plane = i / w / h;
    const int tg = (int)((plane / c) % npart);
    if (tw >= widths[tg]) continue;
    const float v = sym[i] + bias;
    const size_t o = (plane * (h + 2 * pad) + th + pad) * (w + 2 * pad) + tw + pad;
    for (int j = 0; j < rep; j++) ctx[o + j * rep_stride] = v;
  }
}

// entropy_add_cuda.cu:25-44
__global__ __launch_bounds__(kBlock) void entropy_add_kernel(
    float *__restrict__ y, const float *__restrict__ x, const int32_t *__restrict__ order, int lo,
    int len, int nimg, int channel, int cpg, int npart, int h, int w, int pad, int psum) {
  const int total = cpg * len * nimg;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const int pn = i % nimg;
    const int pp = i / nimg;
    const int pb = pp % len;
    const int og = pp / len;
    const Pos p = decode_pos(order[lo + pb], h, w);
    const int tc = psum - p.tw - p.row;
    const int pout = tc * cpg + og;
    const size_t idx =
        (((size_t)(pn * npart + p.tg) * channel + pout) * (h + 2 * pad) + p.th + pad) * (w + 2 * pad) +
        p.tw + pad;
    y[idx] = y[idx] + x[idx];
  }
}

// d_extract_cuda_v2.cu:34-52 / 110-132.  i = (img*len + l)*cpn + ci.
// sections > 0: batch layout, image index split into section = img / nout.
__global__ __launch_bounds__(kBlock) void dextract2_kernel(
    const float *__restrict__ x, float *__restrict__ out, const int32_t *__restrict__ order, int lo,
    int len, int nimg, int channel, int cpn, int npart, int h, int w, int psum, int nout,
    long long section_stride) {
  const int total = len * nimg * cpn;
  const int inner = len * cpn * (nout > 0 ? nout : 1);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const int ci = i % cpn;
    const int tl = (i / cpn) % len;
    const int tn = i / cpn / len;
    const Pos p = decode_pos(order[lo + tl], h, w);
    const int tc = psum - p.tw - p.row;
    const size_t idx = ((((size_t)tn * npart + p.tg) * channel + tc * cpn + ci) * h + p.th) * w + p.tw;
    size_t o = i;
    if (nout > 0) o = (size_t)(i / inner) * section_stride + (i % inner);
    out[o] = x[idx];
  }
}

// Masked grouped 5 x 5 convolution evaluated at wavefront positions only.
//
// Work decomposition: a workgroup owns kPosPerWg positions of ONE plane of ONE
// image, one per wave; they share the output group (tc = psum - plane) and
// therefore the GO x (cin*25) weight rows, which are staged in LDS once per
// workgroup while the gathers are in flight.  Lanes stride over the flattened
// reduction index kk = (kh*5 + kw)*cin + ci (tap-major, channel-minor: the order
// in which the engine's channels-last buffers are contiguous), keep GO partial
// sums and finish with a butterfly.  Reduction order (part of the bitstream
// contract, restated by the oracle): lane l accumulates kk = l, l+64, ... with
// fmaf, skipping taps the causal mask forbids, then
// v += shfl_xor(v, 32, 16, 8, 4, 2, 1).
// All ITER gathers of a lane are issued before the first fmaf (the step is
// latency-bound); weights come from LDS (consecutive lanes, conflict-free).
//
// VHALO: halo taps are computed on the fly from the neighbouring tile's interior
// (lerp of two columns, or the circular wrap) instead of being read from a stored
// halo, which removes the separate halo-update launch.  The value is the one
// EntropyCtxPadRun2 would have stored: its sources are written exactly once and
// the causal mask only lets a tap through after that.
//
// reference: entropy_conv_cuda_v2.cu:326-380 (one 128-thread block per output
// scalar with a warp-32 shuffle tail; not translatable to wave64).
constexpr int kConvBlock = 512;                 // 8 waves, one position each
constexpr int kPosPerWg = kConvBlock / kWave;

struct VHalo {
  const int32_t *widths, *col;
  const float *wgt;
};

template <int GO, int ITER, bool VHALO>
__global__ __launch_bounds__(kConvBlock) void entropy_conv_kernel(
    const float *__restrict__ x, const float *__restrict__ weight, const float *__restrict__ bias,
    const float *__restrict__ slope, const float *__restrict__ residual, float *__restrict__ y,
    const int32_t *__restrict__ order, const int32_t *__restrict__ plane_start, int first_plane,
    int nplane, int chunks, int per_set, int cin, int cout, int group_in, int constrain, int npart, int h,
    int w, int pad_in, int pad_out, int psum, VHalo vh) {
  constexpr int K = 5, KK = 25, HALF = 2;
  extern __shared__ float wl[];  // [GO][red]
  const int chunk = blockIdx.x % chunks;
  const int pl = (blockIdx.x / chunks) % nplane;
  const int pn = blockIdx.x / chunks / nplane;
  const int plane = first_plane + pl;
  const int lo = plane_start[plane];
  const int cnt = plane_start[plane + 1] - lo;
  const int first = chunk * kPosPerWg;
  if (first >= cnt) return;  // uniform for the workgroup
  const int tc = psum - plane;  // output group of every position of this plane
  const int set = pn / per_set;
  const int red = cin * KK;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int pi = first + wave;
  const bool active = pi < cnt;  // wave-uniform
  // the position's schedule entry and the weight rows are independent loads:
  // issue both before anything waits
  const int hw = order[lo + (active ? pi : first)];
  {
    // LDS copy in reduction order: wl[o][tap*cin + ci] = W[o][ci][tap]
    const float *wrow = weight + ((size_t)set * cout + tc * GO) * red;
    for (int i = threadIdx.x; i < GO * red; i += kConvBlock) {
      const int o = i / red, kk = i - o * red;
      wl[i] = wrow[o * red + (kk % cin) * KK + kk / cin];
    }
  }
  const Pos p = decode_pos(hw, h, w);
  const int hin = h + 2 * pad_in, win = w + 2 * pad_in;
  const size_t in_plane = (size_t)hin * win;
  const size_t tile_stride = (size_t)cin * in_plane;
  const int slack = (constrain == 5) ? 0 : 1;
  const int qn = pn * npart + p.tg;
  const float *ximg = x + (size_t)pn * npart * tile_stride;
  float xv[ITER];
  bool ok[ITER];
  bool edge = false;
  int valid = 0;
  if (VHALO) {
    valid = vh.widths[p.tg];
    edge = (p.th < HALF) || (p.th >= h - HALF) || (p.tw + HALF >= valid);
  }
  if (!edge) {
    const float *xin = ximg + (size_t)p.tg * tile_stride + (size_t)(p.th - HALF + pad_in) * win + p.tw - HALF + pad_in;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane + it * kWave;
      const int kc = kk < red ? kk : red - 1;
      const int ci = kc % cin, tap = kc / cin;
      const int kw = tap % K, kh = tap / K;
      // causality: input group g at (qh, pw) is usable iff g + qh + pw < psum
      // (constrain 5) or <= psum (constrain 6); qh + pw = row + tw - 4 + kh + kw
      const int nch = (tc + 2 * HALF - kh - kw + slack) * group_in;
      ok[it] = active && (kk < red) && (ci < nch);
      xv[it] = ok[it] ? xin[(size_t)ci * in_plane + kh * win + kw] : 0.f;
    }
  } else {
    const int rows = h * npart;
    // first round: table entries of the taps that fall into a halo row
    int src_off[ITER], src_off1[ITER];  // element offsets inside the image, -1 = zero
    float src_w[ITER];
#pragma unroll
    for (int it = 0; it < ITER; it++) {
      const int kk = lane + it * kWave;
      const int kc = kk < red ? kk : red - 1;
      const int ci = kc % cin, tap = kc / cin;
      const int kw = tap % K, kh = tap / K;
      const int nch = (tc + 2 * HALF - kh - kw + slack) * group_in;
      ok[it] = active && (kk < red) && (ci < nch);
      src_off[it] = -1;
      src_off1[it] = -1;
      src_w[it] = 1.f;
      if (ok[it]) {
        const int pr = p.th + kh;  // padded coordinates of the tap
        int pc = p.tw + kw;
        if (pc >= valid + pad_in) pc -= valid;  // circular wrap of the first columns
        const int cbase = ci * (int)in_plane;
        if (pr >= pad_in && pr < h + pad_in) {
          src_off[it] = p.tg * (int)tile_stride + cbase + pr * win + pc;  // left halo columns read zeros
        } else if (pc >= pad_in) {
          const int side = pr >= h + pad_in;
          const int r = side ? pr - (h + pad_in) : pr;
          const int row = side ? (p.tg + 1) * h + r : p.tg * h - pad_in + r;
          if (row >= 0 && row < rows) {
            const int e = ((p.tg * 2 + side) * pad_in + r) * w + pc - pad_in;
            const int c = vh.col[e];
            if (c != -2) {
              const int st = row / h;
              const int rbase = st * (int)tile_stride + cbase + (row - st * h + pad_in) * win + pad_in;
              const int wst = vh.widths[st];
              int c1 = c + 1;
              c1 = c1 >= wst ? c1 - wst : c1;
              src_w[it] = vh.wgt[e];
              src_off[it] = (c < 0) ? -1 : rbase + c;
              src_off1[it] = rbase + c1;
            }
          }
        }
      }
    }
    // second round: the values
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
    const int kc = kk < red ? kk : red - 1;
#pragma unroll
    for (int o = 0; o < GO; o++) {
      const float f = fmaf(xv[it], wl[o * red + kc], acc[o]);
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
    const size_t oidx =
        (((size_t)qn * cout + pout) * (h + 2 * pad_out) + p.th + pad_out) * (w + 2 * pad_out) + p.tw + pad_out;
    if (residual) v = v + residual[oidx];  // EntropyAdd folded in (entropy_add_cuda.cu:42)
    y[oidx] = v;
  }
}

// PCONV.EntropyGmmTableOp: parameters in three packed arrays, modified in place
__global__ __launch_bounds__(kBlock) void gmm_table_kernel(float *__restrict__ weight,
                                                           float *__restrict__ delta,
                                                           const float *__restrict__ mean,
                                                           float *__restrict__ table, int tn, int ng,
                                                           int nstep, float bias, float total, float beta,
                                                           int batch_arith) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= tn) return;
  float wt[kMaxGauss], dl[kMaxGauss], mu[kMaxGauss];
  for (int k = 0; k < ng; k++) {
    wt[k] = weight[i * ng + k];
    dl[k] = delta[i * ng + k];
    mu[k] = mean[i * ng + k];
  }
  gmm_prepare_row(wt, dl, ng, beta);
  for (int k = 0; k < ng; k++) {
    weight[i * ng + k] = wt[k];
    delta[i * ng + k] = dl[k];
  }
  gmm_cdf_row<float>(wt, dl, mu, ng, nstep, bias, total, batch_arith, table + (size_t)i * (nstep + 1));
}

// Engine step: DExtract2Batch + EntropyBatchGmmTable (+ DExtract2 of the labels)
// in one launch.  y (3*nimg*npart, ngroup*3, h, w) is the last layer's output;
// replica 0/1/2 = mixture weights / deltas / means.  Row r = img*len + l.
__global__ __launch_bounds__(kBlock) void step_tables_kernel(
    const float *__restrict__ y, const float *__restrict__ symbols, int32_t *__restrict__ table,
    int32_t *__restrict__ labels, const int32_t *__restrict__ order, int lo, int len, int nimg, int ngroup,
    int npart, int h, int w, int psum, int nstep, float bias, float total, float beta) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= len * nimg) return;
  const int l = r % len, n = r / len;
  const Pos p = decode_pos(order[lo + l], h, w);
  const int tc = psum - p.tw - p.row;
  const size_t plane = (size_t)h * w;
  const size_t pix = (size_t)p.th * w + p.tw;
  float par[3][3];
#pragma unroll
  for (int rep = 0; rep < 3; rep++) {
    const float *base = y + (((size_t)(rep * nimg + n) * npart + p.tg) * (ngroup * 3) + tc * 3) * plane + pix;
#pragma unroll
    for (int k = 0; k < 3; k++) par[rep][k] = base[k * plane];
  }
  gmm_prepare_row(par[0], par[1], 3, beta);
  gmm_cdf_row<int32_t>(par[0], par[1], par[2], 3, nstep, bias, total, 1, table + (size_t)r * (nstep + 1));
  if (symbols)
    labels[r] = (int32_t)symbols[(((size_t)n * npart + p.tg) * ngroup + tc) * plane + pix];
}

}  // namespace

extern "C" int pconv_dinput2(const float *packed, float *ctx, const int32_t *order, int lo, int len,
                             int nimg, int ngroup, int npart, int h, int w, int pad, int psum,
                             float bias, int rep, void *stream) {
  PCONV_REQUIRE(packed && ctx && order, "dinput2: null pointer");
  if (len <= 0 || nimg <= 0) return PCONV_OK;
  hipLaunchKernelGGL(dinput2_kernel, dim3(pconv_grid((long long)len * nimg)), dim3(kBlock), 0,
                     as_stream(stream), packed, ctx, order, lo, len, nimg, ngroup, npart, h, w, pad,
                     psum, bias, rep);
  PCONV_LAUNCH_CHECK("dinput2");
  return PCONV_OK;
}

extern "C" int pconv_ctx_to_symbols(const float *ctx, float *out, const int32_t *widths, int tn, int c,
                                    int h, int w, int pad, int npart, float bias, void *stream) {
  PCONV_REQUIRE(ctx && out && widths && tn > 0 && c > 0, "ctx_to_symbols: bad argument");
  const long long total = (long long)tn * c * h * w;
  hipLaunchKernelGGL(ctx_to_symbols_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream), ctx,
                     out, widths, c, h, w, pad, npart, bias, total);
  PCONV_LAUNCH_CHECK("ctx_to_symbols");
  return PCONV_OK;
}

extern "C" int pconv_symbols_to_ctx(const float *symbols, float *ctx, const int32_t *widths, int tn, int c,
                                    int h, int w, int pad, int npart, float bias, int rep, void *stream) {
  PCONV_REQUIRE(symbols && ctx && widths && tn > 0 && c > 0 && rep > 0, "symbols_to_ctx: bad argument");
  const long long total = (long long)tn * c * h * w;
  hipLaunchKernelGGL(symbols_to_ctx_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream),
                     symbols, ctx, widths, c, h, w, pad, npart, bias, rep, total);
  PCONV_LAUNCH_CHECK("symbols_to_ctx");
  return PCONV_OK;
}

extern "C" int pconv_ctx_pad_run2(float *data, const int32_t *dst, const int32_t *src0,
                                  const int32_t *src1, const float *wgt, const int32_t *entry_plane,
                                  int lo, int len, int nimg, int cpn, int channel, int npart, int h,
                                  int w, int pad, int psum, void *stream) {
  PCONV_REQUIRE(data && dst && src0 && src1 && wgt && entry_plane, "ctx_pad_run2: null pointer");
  if (len <= 0) return PCONV_OK;
  const size_t plane_sz = (size_t)(h + 2 * pad) * (w + 2 * pad);
  const size_t img_stride = plane_sz * channel * npart;
  hipLaunchKernelGGL(ctx_pad_run2_kernel, dim3(pconv_grid((long long)nimg * cpn * len)),
                     dim3(kBlock), 0, as_stream(stream), data, dst, src0, src1, wgt, entry_plane, lo,
                     len, nimg, cpn, plane_sz, img_stride, psum);
  PCONV_LAUNCH_CHECK("ctx_pad_run2");
  return PCONV_OK;
}

extern "C" int pconv_entropy_conv(const float *x, const float *weight, const float *bias,
                                  const float *slope, float *y, const int32_t *order,
                                  const int32_t *plane_start, int first_plane, int nplane,
                                  int max_plane_len, int nimg, int per_set, int cin, int cout, int ngroup,
                                  int k, int constrain, int npart, int h, int w, int pad_in,
                                  int pad_out, int psum, const float *residual, const int32_t *widths,
                                  const int32_t *vh_col, const float *vh_wgt, void *stream) {
  PCONV_REQUIRE(x && weight && bias && y && order && plane_start, "entropy_conv: null pointer");
  PCONV_REQUIRE(ngroup > 0 && cin % ngroup == 0 && cout % ngroup == 0 && per_set > 0,
                "entropy_conv: bad channel grouping");
  PCONV_REQUIRE(constrain == 5 || constrain == 6, "entropy_conv: constrain must be 5 or 6");
  PCONV_REQUIRE(k == 5, "entropy_conv: kernel size %d not supported (5)", k);
  PCONV_REQUIRE(pad_in >= k / 2, "entropy_conv: pad_in %d smaller than half kernel %d", pad_in, k / 2);
  PCONV_REQUIRE(!vh_col || (widths && vh_wgt && pad_in == k / 2), "entropy_conv: incomplete virtual-halo tables");
  if (nplane <= 0 || nimg <= 0 || max_plane_len <= 0) return PCONV_OK;
  const int go = cout / ngroup;
  const int red = cin * k * k;
  const int iter = (red + kWave - 1) / kWave;
  const int chunks = (max_plane_len + kPosPerWg - 1) / kPosPerWg;
  const long long grid = (long long)nimg * nplane * chunks;
  PCONV_REQUIRE(grid < (1LL << 31), "entropy_conv: grid too large");
  const size_t smem = (size_t)go * red * sizeof(float);
  PCONV_REQUIRE(smem <= 64 * 1024, "entropy_conv: weight rows do not fit LDS");
  VHalo vh = {widths, vh_col, vh_wgt};
#define LAUNCH_CONV(GO, ITER, VH)                                                                        \
  hipLaunchKernelGGL((entropy_conv_kernel<GO, ITER, VH>), dim3((unsigned)grid), dim3(kConvBlock), smem,       \
                     as_stream(stream), x, weight, bias, slope, residual, y, order, plane_start,          \
                     first_plane, nplane, chunks, per_set, cin, cout, cin / ngroup, constrain, npart, h,  \
                     w, pad_in, pad_out, psum, vh)
#define BY_VH(GO, ITER)          \
  if (vh_col)                    \
    LAUNCH_CONV(GO, ITER, true); \
  else                           \
    LAUNCH_CONV(GO, ITER, false)
#define BY_ITER(GO)                                                                          \
  if (iter <= 2) { BY_VH(GO, 2); }                                                           \
  else if (iter <= 6) { BY_VH(GO, 6); }                                                      \
  else if (iter <= 12) { BY_VH(GO, 12); }                                                    \
  else if (iter <= 17) { BY_VH(GO, 17); }                                                    \
  else if (iter <= 24) { BY_VH(GO, 24); }                                                    \
  else if (iter <= 33) { BY_VH(GO, 33); }  /* 28 groups x 3: valid_dim 112 */                \
  else if (iter <= 57) { BY_VH(GO, 57); }  /* 48 groups x 3: valid_dim 192 */                \
  else {                                                                                     \
    pconv_set_error("entropy_conv: reduction length %d too long (max 3648)", red);           \
    return PCONV_EINVAL;                                                                     \
  }
  switch (go) {
    case 1: BY_ITER(1); break;
    case 2: BY_ITER(2); break;
    case 3: BY_ITER(3); break;
    case 4: BY_ITER(4); break;
    default:
      pconv_set_error("entropy_conv: %d outputs per group not supported (1..4)", go);
      return PCONV_EINVAL;
  }
#undef BY_ITER
#undef BY_VH
#undef LAUNCH_CONV
  PCONV_LAUNCH_CHECK("entropy_conv");
  return PCONV_OK;
}

extern "C" int pconv_entropy_add(float *y, const float *x, const int32_t *order, int lo, int len,
                                 int nimg, int channel, int ngroup, int npart, int h, int w,
                                 int pad, int psum, void *stream) {
  PCONV_REQUIRE(y && x && order && ngroup > 0, "entropy_add: bad argument");
  if (len <= 0 || nimg <= 0) return PCONV_OK;
  const int cpg = channel / ngroup;
  hipLaunchKernelGGL(entropy_add_kernel, dim3(pconv_grid((long long)cpg * len * nimg)),
                     dim3(kBlock), 0, as_stream(stream), y, x, order, lo, len, nimg, channel, cpg,
                     npart, h, w, pad, psum);
  PCONV_LAUNCH_CHECK("entropy_add");
  return PCONV_OK;
}

extern "C" int pconv_dextract2(const float *x, float *out, const int32_t *order, int lo, int len,
                               int nimg, int channel, int cpn, int npart, int h, int w, int psum,
                               void *stream) {
  PCONV_REQUIRE(x && out && order, "dextract2: null pointer");
  if (len <= 0 || nimg <= 0) return PCONV_OK;
  hipLaunchKernelGGL(dextract2_kernel, dim3(pconv_grid((long long)len * nimg * cpn)), dim3(kBlock),
                     0, as_stream(stream), x, out, order, lo, len, nimg, channel, cpn, npart, h, w,
                     psum, 0, 0LL);
  PCONV_LAUNCH_CHECK("dextract2");
  return PCONV_OK;
}

extern "C" int pconv_dextract2_batch(const float *x, float *out, const int32_t *order, int lo,
                                     int len, int nimg, int channel, int cpn, int npart, int h,
                                     int w, int psum, int nout, long long section_stride,
                                     void *stream) {
  PCONV_REQUIRE(x && out && order && nout > 0, "dextract2_batch: bad argument");
  if (len <= 0 || nimg <= 0) return PCONV_OK;
  hipLaunchKernelGGL(dextract2_kernel, dim3(pconv_grid((long long)len * nimg * cpn)), dim3(kBlock),
                     0, as_stream(stream), x, out, order, lo, len, nimg, channel, cpn, npart, h, w,
                     psum, nout, section_stride);
  PCONV_LAUNCH_CHECK("dextract2_batch");
  return PCONV_OK;
}

extern "C" int pconv_gmm_table(float *weight, float *delta, const float *mean, float *table,
                               int tn, int ng, int nstep, float bias, float total, float beta,
                               int batch_arith, void *stream) {
  PCONV_REQUIRE(weight && delta && mean && table, "gmm_table: null pointer");
  PCONV_REQUIRE(ng > 0 && ng <= kMaxGauss && nstep > 0, "gmm_table: bad ng/nstep");
  if (tn <= 0) return PCONV_OK;
  const unsigned grid = (tn + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(gmm_table_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), weight, delta, mean,
                     table, tn, ng, nstep, bias, total, beta, batch_arith);
  PCONV_LAUNCH_CHECK("gmm_table");
  return PCONV_OK;
}

extern "C" int pconv_step_tables(const float *y, const float *symbols, int32_t *table, int32_t *labels,
                                 const int32_t *order, int lo, int len, int nimg, int ngroup, int npart,
                                 int h, int w, int psum, int nstep, float bias, float total, float beta,
                                 void *stream) {
  PCONV_REQUIRE(y && table && order, "step_tables: null pointer");
  PCONV_REQUIRE(!symbols || labels, "step_tables: symbols given without a label buffer");
  if (len <= 0 || nimg <= 0) return PCONV_OK;
  const unsigned grid = (len * nimg + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(step_tables_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), y, symbols, table,
                     labels, order, lo, len, nimg, ngroup, npart, h, w, psum, nstep, bias, total, beta);
  PCONV_LAUNCH_CHECK("step_tables");
  return PCONV_OK;
}
