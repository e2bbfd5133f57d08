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

// Masked grouped k x k convolution evaluated at wavefront positions only.
// One wave per (image, position); lanes stride over the flattened reduction
// index kk = (ci*k + kh)*k + kw (the weight's own memory order, so weight loads
// are contiguous across lanes) and keep GO partial sums, one per output channel
// of the group.  Reduction order (part of the bitstream contract, restated by
// the oracle): lane l accumulates kk = l, l+64, ... with fmaf, then a butterfly
// v += shfl_xor(v, 32, 16, 8, 4, 2, 1).
// reference: entropy_conv_cuda_v2.cu:326-380 (one 128-thread block per output
// scalar with a warp-32 shuffle tail; not translatable to wave64).
template <int GO>
__global__ __launch_bounds__(kBlock) void entropy_conv_kernel(
    const float *__restrict__ x, const float *__restrict__ weight, const float *__restrict__ bias,
    const float *__restrict__ slope, float *__restrict__ y, const int32_t *__restrict__ order,
    int lo, int len, int nimg, int per_set, int cin, int cout, int group_in, int k, int constrain,
    int npart, int h, int w, int pad_in, int pad_out, int psum) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = (blockIdx.x * kBlock + threadIdx.x) / kWave;
  const int nwave = gridDim.x * (kBlock / kWave);
  const int half = k / 2;
  const int kk_sz = k * k;
  const int hin = h + 2 * pad_in, win = w + 2 * pad_in;
  const size_t in_plane = (size_t)hin * win;
  const int red = cin * kk_sz;
  for (int item = wave; item < len * nimg; item += nwave) {
    const int pb = item % len;
    const int pn = item / len;
    const Pos p = decode_pos(order[lo + pb], h, w);
    const int tc = psum - p.tw - p.row;  // output group handled at this position
    const int set = pn / per_set;
    const int qn = pn * npart + p.tg;
    const float *xin = x + (size_t)qn * cin * in_plane;
    const float *wrow = weight + ((size_t)set * cout + tc * GO) * red;
    float acc[GO];
#pragma unroll
    for (int o = 0; o < GO; o++) acc[o] = 0.f;
    for (int kk = lane; kk < red; kk += kWave) {
      const int kw = kk % k;
      const int kh = (kk / k) % k;
      const int ci = kk / kk_sz;
      // causality: input group g at (qh, pw) is usable iff g + qh + pw < psum
      // (constrain 5) or <= psum (constrain 6)
      const int qh = p.row - half + kh;
      const int pw = p.tw - half + kw;
      int nch = (constrain == 5 ? (psum - qh - pw) : (psum - qh - pw + 1)) * group_in;
      if (ci < nch) {
        const float v = xin[(size_t)ci * in_plane + (size_t)(p.th - half + kh + pad_in) * win + pw + pad_in];
#pragma unroll
        for (int o = 0; o < GO; o++) acc[o] = fmaf(v, wrow[(size_t)o * red + kk], acc[o]);
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
      y[(((size_t)qn * cout + pout) * (h + 2 * pad_out) + p.th + pad_out) * (w + 2 * pad_out) + p.tw +
        pad_out] = v;
    }
  }
}

// entropy_gmm_table_cuda.cu:29-57 (softmax, relu+beta in place)
__global__ __launch_bounds__(kBlock) void gmm_prepare_kernel(float *__restrict__ weight,
                                                             float *__restrict__ delta, int tn,
                                                             int ng, float beta) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= tn) return;
  float tmp[16];
  float mval = -1e10, psum = 0;
  for (int k = 0; k < ng; k++) {
    tmp[k] = weight[i * ng + k];
    if (mval < tmp[k]) mval = tmp[k];
  }
  for (int k = 0; k < ng; k++) {
    tmp[k] = pconv_expf(tmp[k] - mval);
    psum += tmp[k];
  }
  for (int k = 0; k < ng; k++) {
    weight[i * ng + k] = tmp[k] / psum;
    float d = delta[i * ng + k];
    delta[i * ng + k] = d < 0 ? beta : d + beta;
  }
}

// CDF rows + monotonicity repair, one thread per row
// (entropy_gmm_table_cuda.cu:83-105,136-153)
__global__ __launch_bounds__(kBlock) void gmm_table_kernel(const float *__restrict__ weight,
                                                           const float *__restrict__ delta,
                                                           const float *__restrict__ mean,
                                                           float *__restrict__ table, int tn, int ng,
                                                           int nstep, float bias, float total,
                                                           int batch_arith) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= tn) return;
  const float s2 = 1. / sqrt(2.0);
  float *row = table + (size_t)i * (nstep + 1);
  float prev = 0.f, shift = 0.f, widest = 0.f;
  int widest_at = 0;
  row[0] = 0.f;
  // first pass: raw integer CDF with the running +1 repair applied on the fly
  for (int pt = 1; pt <= nstep; pt++) {
    float cur;
    if (pt == nstep) {
      cur = static_cast<int>(total);
    } else {
      float v = pt - 1 - bias + 0.5, ps = 0;
      for (int k = 0; k < ng; k++) {
        const float e = pconv_erff(s2 * (v - mean[i * ng + k]) / delta[i * ng + k]);
        if (batch_arith) {
          ps = ps + weight[i * ng + k] * (0.5 + 0.5 * e);  // double inside, as :148
        } else {
          const float f = 0.5 + 0.5 * e;  // rounded to float, as :72-73
          ps = ps + weight[i * ng + k] * f;
        }
      }
      cur = static_cast<int>(total * ps + 0.5);
    }
    // check kernel: compares the raw entry with the already shifted previous one
    if (cur <= prev) shift += 1;
    cur += shift;
    if (cur - prev > widest) {
      widest = cur - prev;
      widest_at = pt - 1;
    }
    row[pt] = cur;
    prev = cur;
  }
  if (shift > 0)
    for (int pt = widest_at; pt < nstep; pt++) row[pt + 1] -= shift;
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
                                  const float *slope, float *y, const int32_t *order, int lo,
                                  int len, int nimg, int per_set, int cin, int cout, int ngroup,
                                  int k, int constrain, int npart, int h, int w, int pad_in,
                                  int pad_out, int psum, void *stream) {
  PCONV_REQUIRE(x && weight && bias && y && order, "entropy_conv: null pointer");
  PCONV_REQUIRE(ngroup > 0 && cin % ngroup == 0 && cout % ngroup == 0 && per_set > 0,
                "entropy_conv: bad channel grouping");
  PCONV_REQUIRE(constrain == 5 || constrain == 6, "entropy_conv: constrain must be 5 or 6");
  PCONV_REQUIRE(pad_in >= k / 2, "entropy_conv: pad_in %d smaller than half kernel %d", pad_in, k / 2);
  if (len <= 0 || nimg <= 0) return PCONV_OK;
  const int go = cout / ngroup;
  const long long waves = (long long)len * nimg;
  const unsigned grid = pconv_grid(waves * kWave);
#define LAUNCH_CONV(GO)                                                                          \
  hipLaunchKernelGGL(entropy_conv_kernel<GO>, dim3(grid), dim3(kBlock), 0, as_stream(stream), x, \
                     weight, bias, slope, y, order, lo, len, nimg, per_set, cin, cout,           \
                     cin / ngroup, k, constrain, npart, h, w, pad_in, pad_out, psum)
  switch (go) {
    case 1: LAUNCH_CONV(1); break;
    case 2: LAUNCH_CONV(2); break;
    case 3: LAUNCH_CONV(3); break;
    case 4: LAUNCH_CONV(4); break;
    case 6: LAUNCH_CONV(6); break;
    case 8: LAUNCH_CONV(8); break;
    default:
      pconv_set_error("entropy_conv: %d outputs per group not supported (1,2,3,4,6,8)", go);
      return PCONV_EINVAL;
  }
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
  PCONV_REQUIRE(ng > 0 && ng <= 16 && nstep > 0, "gmm_table: bad ng/nstep");
  if (tn <= 0) return PCONV_OK;
  const unsigned grid = (tn + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(gmm_prepare_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), weight,
                     delta, tn, ng, beta);
  hipLaunchKernelGGL(gmm_table_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), weight, delta,
                     mean, table, tn, ng, nstep, bias, total, batch_arith);
  PCONV_LAUNCH_CHECK("gmm_table");
  return PCONV_OK;
}
