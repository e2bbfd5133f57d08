// Element-wise ops of the codec path: quantiser / de-quantiser, viewport
// projection, ContextReshape, MaskConstrain, EntropyGmm loss.
#include "common.h"
#include "../../include/pconv_detmath.h"

namespace {

constexpr int kBlock = 256;

// level table of the quantiser: tab[c,0] = w[c,0], tab[c,j] = exp(w[c,j])
// (pseudo_quant_cuda.cu:37-45)
__global__ void quant_levels_kernel(const float *__restrict__ w, float *__restrict__ tab, int n,
                                    int levels) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  tab[i] = (i % levels == 0) ? w[i] : pconv_expf(w[i]);
}

// cumulative level table of the de-quantiser (pseudo_dquant_cuda.cu:24-32)
__global__ void dquant_levels_kernel(const float *__restrict__ w, float *__restrict__ tab, int nch,
                                     int levels) {
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= nch) return;
  float acc = w[ch * levels];
  tab[ch * levels] = acc;
  for (int j = 1; j < levels; j++) {
    acc = acc + pconv_expf(w[ch * levels + j]);
    tab[ch * levels + j] = acc;
  }
}

// pseudo_quant_cuda.cu:48-85
__global__ __launch_bounds__(kBlock) void quant_kernel(
    const float *__restrict__ x, const float *__restrict__ tab, float *__restrict__ out_val,
    float *__restrict__ out_idx, float *__restrict__ count, const int32_t *__restrict__ widths,
    int c, int hw, int w, int levels, int npart, long long total) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int pw = (int)(i % w);
    const long long plane = i / hw;
    const int pc = (int)(plane % c);
    const int pg = (int)((plane / c) % npart);
    float val = 0.f;
    int q = 0;
    if (pw < widths[pg]) {
      const float *lv = tab + pc * levels;
      const float v = x[i];
      float tmp = v - lv[0];
      if (tmp < 0) {
        q = 0;
        val = lv[0];
      } else {
        int j = 1;
        for (; j < levels; j++) {
          tmp -= lv[j];
          if (tmp < 0) break;
        }
        if (j == levels) j--;
        if (tmp + tmp + lv[j] < 0) {
          tmp = tmp + lv[j];
          j--;
        }
        val = v - tmp;
        q = j;
      }
    }
    if (count) {
      // per-call histogram (pseudo_quant_cuda.cu:64,83: atomicAdd(count + pc*levels + j, -1.0)).
      // The sums are integers far below 2^24, so any order gives the same bits: the lanes
      // of a wave that hit the same (channel, level) add once, with their number.
      const bool live = pw < widths[pg];
      const int slot = live ? pc * levels + q : -1;
      const int first = __builtin_amdgcn_readfirstlane(pc);
      if (__all(pc == first)) {
        for (int lv = 0; lv < levels; lv++) {
          const unsigned long long hit = __ballot(live && q == lv);
          if (hit && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(hit))
            atomicAdd(count + first * levels + lv, -(float)__popcll(hit));
        }
      } else if (slot >= 0) {
        atomicAdd(count + slot, -1.0f);
      }
    }
    out_val[i] = val;
    if (out_idx) out_idx[i] = (float)q;
  }
}

// The same quantiser for the codec's shape -- 8 levels, no histogram (eval mode), rows of a multiple of 4 columns --
// with 16-byte accesses (r6): a thread owns 4 consecutive columns of one (tile, channel, row), loads the channel's 8
// level increments ONCE as two 16-byte pieces (the scalar kernel walked them through up to 8 dependent loads per
// element) and runs, per element, the very operations of quant_kernel in the same order: identical bits.
__device__ __forceinline__ void quant8_one(float v, const float (&lv)[8], float &val, float &idx) {
  float tmp = v - lv[0];
  if (tmp < 0) {
    val = lv[0];
    idx = 0.f;
    return;
  }
  int j = 8;
  bool brk = false;
#pragma unroll
  for (int k = 1; k < 8; k++) {
    if (!brk) {
      tmp -= lv[k];
      if (tmp < 0) {
        brk = true;
        j = k;
      }
    }
  }
  if (!brk) j = 7;
  float lvj = lv[1];
#pragma unroll
  for (int k = 2; k < 8; k++) lvj = j == k ? lv[k] : lvj;
  if (tmp + tmp + lvj < 0) {
    tmp = tmp + lvj;
    j--;
  }
  val = v - tmp;
  idx = (float)j;
}

__global__ __launch_bounds__(kBlock) void quant8x4_kernel(const float4 *__restrict__ x, const float *__restrict__ tab,
                                                         float4 *__restrict__ out_val, float4 *__restrict__ out_idx,
                                                         const int32_t *__restrict__ widths, int c, int hw, int w,
                                                         int npart, long long total4) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (long long)gridDim.x * kBlock) {
    const long long e = 4 * i;
    const int pw = (int)(e % w);
    const long long plane = e / hw;
    const int pc = (int)(plane % c);
    const int pg = (int)((plane / c) % npart);
    const int live = widths[pg] - pw;  // columns of this quad inside the tile's valid width
    float4 val = {0.f, 0.f, 0.f, 0.f}, idx = {0.f, 0.f, 0.f, 0.f};
    if (live > 0) {
      const float4 t0 = *reinterpret_cast<const float4 *>(tab + pc * 8), t1 = *reinterpret_cast<const float4 *>(tab + pc * 8 + 4);
      const float lv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      const float4 v = x[i];
      quant8_one(v.x, lv, val.x, idx.x);
      if (live > 1) quant8_one(v.y, lv, val.y, idx.y);
      if (live > 2) quant8_one(v.z, lv, val.z, idx.z);
      if (live > 3) quant8_one(v.w, lv, val.w, idx.w);
    }
    out_val[i] = val;
    if (out_idx) out_idx[i] = idx;
  }
}

__global__ __launch_bounds__(kBlock) void dquant8x4_kernel(const float4 *__restrict__ x, const float *__restrict__ tab,
                                                          float4 *__restrict__ out, const int32_t *__restrict__ widths,
                                                          int c, int hw, int w, int npart, long long total4) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (long long)gridDim.x * kBlock) {
    const long long e = 4 * i;
    const int pw = (int)(e % w);
    const long long plane = e / hw;
    const int pc = (int)(plane % c);
    const int pg = (int)((plane / c) % npart);
    const int live = widths[pg] - pw;
    float4 o = {0.f, 0.f, 0.f, 0.f};
    if (live > 0) {
      const float4 t0 = *reinterpret_cast<const float4 *>(tab + pc * 8), t1 = *reinterpret_cast<const float4 *>(tab + pc * 8 + 4);
      const float lv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      const float4 v = x[i];
      auto look = [&](float s) {
        const int k = static_cast<int>(s + 0.00001);
        float r = lv[0];
#pragma unroll
        for (int q = 1; q < 8; q++) r = k == q ? lv[q] : r;
        return r;
      };
      o.x = look(v.x);
      if (live > 1) o.y = look(v.y);
      if (live > 2) o.z = look(v.z);
      if (live > 3) o.w = look(v.w);
    }
    out[i] = o;
  }
}

// pseudo_dquant_cuda.cu:34-47
__global__ __launch_bounds__(kBlock) void dquant_kernel(const float *__restrict__ x,
                                                        const float *__restrict__ tab,
                                                        float *__restrict__ out,
                                                        const int32_t *__restrict__ widths, int c,
                                                        int hw, int w, int levels, int npart,
                                                        long long total) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int pw = (int)(i % w);
    const long long plane = i / hw;
    const int pc = (int)(plane % c);
    const int pg = (int)((plane / c) % npart);
    float v = 0.f;
    if (pw < widths[pg]) {
      const int idx = static_cast<int>(x[i] + 0.00001);
      v = tab[pc * levels + idx];
    }
    out[i] = v;
  }
}

// projects_cuda.cu:181-213.  index = (view*nc + plane)*inner + pixel
__global__ __launch_bounds__(kBlock) void project_kernel(const float *__restrict__ in,
                                                         const float *__restrict__ tf,
                                                         float *__restrict__ out, int inner, int hs,
                                                         int ws, int nc, int nearest,
                                                         long long total) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int ps = (int)(i % inner);
    const long long rest = i / inner;
    const int tn = (int)(rest % nc);
    const int tb = (int)(rest / nc);
    const float fx = tf[((size_t)tb * inner + ps) * 2];
    const float fy = tf[((size_t)tb * inner + ps) * 2 + 1];
    const float *img = in + (size_t)tn * hs * ws;
    if (nearest) {
      int tw = static_cast<int>(floor(fx + 0.5)) % ws;
      int th = static_cast<int>(floor(fy + 0.5));
      th = th >= hs ? hs - 1 : th;
      out[i] = img[(size_t)th * ws + tw];
    } else {
      const int tw = static_cast<int>(floorf(fx));
      const int th = static_cast<int>(floorf(fy));
      const int pw = (tw + 1) % ws;
      const int ph = th + 1 >= hs ? hs - 1 : th + 1;
      const float tx = fx - tw;
      const float ty = fy - th;
      const float ntx = 1. - tx;
      const float nty = 1. - ty;
      out[i] = img[(size_t)th * ws + tw] * ntx * nty + img[(size_t)th * ws + pw] * tx * nty +
               img[(size_t)ph * ws + tw] * ntx * ty + img[(size_t)ph * ws + pw] * tx * ty;
    }
  }
}

// context_reshape_cuda.cu:30-39, output-ordered so stores coalesce
__global__ __launch_bounds__(kBlock) void context_reshape_kernel(const float *__restrict__ in,
                                                                 float *__restrict__ out, int inner,
                                                                 int c, int cpg, long long total) {
  for (long long o = (long long)blockIdx.x * kBlock + threadIdx.x; o < total;
       o += (long long)gridDim.x * kBlock) {
    const int k = (int)(o % cpg);
    long long rest = o / cpg;
    const int ps = (int)(rest % inner);
    rest /= inner;
    const int g = (int)(rest % (c / cpg));
    const long long pn = rest / (c / cpg);
    out[o] = in[((size_t)pn * c + g * cpg + k) * inner + ps];
  }
}

// mask_constrain_cuda.cu:19-88
__global__ __launch_bounds__(kBlock) void mask_constrain_kernel(float *__restrict__ w, int cin,
                                                                int sz, int group_in, int group_out,
                                                                int constrain, int total) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int tw = i % sz;
  const int th = (i / sz) % sz;
  const int tc = (i / sz / sz) % cin / group_in;
  const int tn = i / sz / sz / cin / group_out;
  bool zero = false;
  if (constrain == 5) {
    zero = (tw + th + tc >= tn + sz - 1);
  } else if (constrain == 1 || constrain == 2) {
    if (tn > tc) {
      zero = false;
    } else if (tn == tc) {
      if (th < sz / 2)
        zero = false;
      else if (th == sz / 2)
        zero = (constrain == 1) ? !(tw < sz / 2) : !(tw <= sz / 2);
      else
        zero = true;
    } else {
      zero = true;
    }
  } else {
    zero = (tw + th + tc > tn + sz - 1);
  }
  if (zero) w[i] = 0.f;
}

// entropy_gmm_cuda.cu:36-69
__global__ __launch_bounds__(kBlock) void gmm_loss_kernel(
    const float *__restrict__ weight, const float *__restrict__ delta,
    const float *__restrict__ mean, const float *__restrict__ label, float *__restrict__ loss,
    float *__restrict__ d_weight, float *__restrict__ d_delta, float *__restrict__ d_mean,
    float *__restrict__ d_label, int m, int ng) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const float s2 = 1. / sqrtf(2.0f);
  const float sp2 = 1. / sqrt(2. * 3.14159265358979323846);
  float sum_p = 0;
  float dl = 0;
  for (int k = 0; k < ng; k++) {
    const float wk = weight[i * ng + k];
    float xa = label[i] - 0.5 - mean[i * ng + k];
    float xb = label[i] + 0.5 - mean[i * ng + k];
    float id = 1. / delta[i * ng + k];
    float fa = 0.5 + 0.5 * pconv_erff(xa * id * s2);
    float fb = 0.5 + 0.5 * pconv_erff(xb * id * s2);
    float p = fb - fa;
    sum_p = sum_p + wk * p;
    if (d_weight) {
      float ga = sp2 * id * pconv_expf(-0.5 * xa * xa * id * id);
      float gb = sp2 * id * pconv_expf(-0.5 * xb * xb * id * id);
      dl += (gb - ga) * wk;
      d_delta[i * ng + k] = id * (-xb * gb + xa * ga) * wk;
      d_mean[i * ng + k] = (ga - gb) * wk;
      d_weight[i * ng + k] = p;
    }
  }
  loss[i] = -logf(sum_p + 0.0000001);
  if (d_weight) {
    float ip = -1. / (sum_p + 0.0000001);
    d_label[i] = dl * ip;
    for (int k = 0; k < ng; k++) {
      d_delta[i * ng + k] *= ip;
      d_mean[i * ng + k] *= ip;
      d_weight[i * ng + k] *= ip;
    }
  }
}

// ProjectsOp.backward (projects_cuda.cu:257-329): the transpose of the viewport sampling, with
// the same float atomics as the reference; `count` receives the sampling weights
__global__ __launch_bounds__(kBlock) void project_backward_kernel(const float *__restrict__ gout,
                                                                  const float *__restrict__ tf,
                                                                  float *__restrict__ gin, float *__restrict__ count,
                                                                  int inner, int hs, int ws, int nc, int nearest,
                                                                  long long total) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int ps = (int)(i % inner);
    const long long rest = i / inner;
    const int tn = (int)(rest % nc);
    const int tb = (int)(rest / nc);
    const float fx = tf[((size_t)tb * inner + ps) * 2];
    const float fy = tf[((size_t)tb * inner + ps) * 2 + 1];
    float *img = gin + (size_t)tn * hs * ws;
    float *cnt = count + (size_t)tn * hs * ws;
    const float g = gout[i];
    if (nearest) {
      int tw = static_cast<int>(floor(fx + 0.5)) % ws;
      int th = static_cast<int>(floor(fy + 0.5));
      th = th >= hs ? hs - 1 : th;
      atomicAdd(img + (size_t)th * ws + tw, g);
      atomicAdd(cnt + (size_t)th * ws + tw, 1.f);
    } else {
      const int tw = static_cast<int>(floorf(fx));
      const int th = static_cast<int>(floorf(fy));
      const int pw = (tw + 1) % ws;
      const int ph = th + 1 >= hs ? hs - 1 : th + 1;
      const float tx = fx - tw;
      const float ty = fy - th;
      const float ntx = 1. - tx;
      const float nty = 1. - ty;
      atomicAdd(img + (size_t)th * ws + tw, ntx * nty * g);
      atomicAdd(cnt + (size_t)th * ws + tw, ntx * nty);
      atomicAdd(img + (size_t)th * ws + pw, tx * nty * g);
      atomicAdd(cnt + (size_t)th * ws + pw, tx * nty);
      atomicAdd(img + (size_t)ph * ws + tw, ntx * ty * g);
      atomicAdd(cnt + (size_t)ph * ws + tw, ntx * ty);
      atomicAdd(img + (size_t)ph * ws + pw, tx * ty * g);
      atomicAdd(cnt + (size_t)ph * ws + pw, tx * ty);
    }
  }
}

// PseudoQuantOp.backward (pseudo_quant_cuda.cu:197-311), one pass over the activation:
//   g_in = g_val + alpha * g_idx / beta   (straight-through for the value, the index output's
//          gradient scaled by the width of the level interval the input sits in), zero in the
//          dead columns;
//   bins[c][q] += val - x  for the element's level q (the reference adds it to levels 0..q with
//          q + 1 atomics; the suffix sum over q is taken by quant_weight_grad_kernel below).
__global__ __launch_bounds__(kBlock) void quant_backward_kernel(
    const float *__restrict__ x, const float *__restrict__ val, const float *__restrict__ idx,
    const float *__restrict__ g_val, const float *__restrict__ g_idx, const float *__restrict__ tab,
    float *__restrict__ g_in, float *__restrict__ bins, const int32_t *__restrict__ widths, float alpha, int c,
    int hw, int w, int levels, int npart, long long total) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (long long)gridDim.x * kBlock) {
    const int pw = (int)(i % w);
    const long long plane = i / hw;
    const int pc = (int)(plane % c);
    const int pg = (int)((plane / c) % npart);
    float g = 0.f;
    if (pw < widths[pg]) {
      const float xv = x[i], tv = val[i];
      const int q = (int)idx[i];
      atomicAdd(bins + pc * levels + q, tv - xv);
      g = g_val[i];
      if (g_idx) {
        const float *lv = tab + pc * levels;
        float beta;
        if (tv < xv) {
          beta = q < levels - 1 ? lv[q + 1] : 10000.f;
        } else if (tv > xv) {
          beta = q > 0 ? lv[q] : 10000.f;
        } else if (q == 0) {
          beta = lv[1];
        } else if (q < levels - 1) {
          beta = (lv[q] + lv[q + 1]) / 2.0;
        } else {
          beta = lv[q];
        }
        if (beta < 0.001) beta = 0.001;
        g = g + alpha * g_idx[i] / beta;
      }
    }
    g_in[i] = g;
  }
}

// weight_diff[c][j] = sum over levels q >= j of bins[c][q], times d level / d weight
// (exp(w) = the level table entry for j > 0, 1 for j = 0) (pseudo_quant_cuda.cu:205-233)
__global__ void quant_weight_grad_kernel(const float *__restrict__ bins, const float *__restrict__ tab,
                                         float *__restrict__ g_w, int nch, int levels) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= nch) return;
  float acc = 0.f;
  for (int j = levels - 1; j >= 0; j--) {
    acc += bins[ch * levels + j];
    g_w[ch * levels + j] = j ? acc * tab[ch * levels + j] : acc;
  }
}

}  // namespace

extern "C" int pconv_quant(const float *x, const float *weight, float *level_tab, float *out_val,
                           float *out_idx, float *count, const int32_t *widths, int tn, int c,
                           int h, int w, int levels, int npart, void *stream) {
  PCONV_REQUIRE(x && weight && level_tab && out_val && widths, "quant: null pointer");
  PCONV_REQUIRE(tn > 0 && c > 0 && h > 0 && w > 0 && levels > 1, "quant: bad shape");
  const int ntab = c * levels;
  hipLaunchKernelGGL(quant_levels_kernel, dim3((ntab + 255) / 256), dim3(256), 0,
                     as_stream(stream), weight, level_tab, ntab, levels);
  const long long total = (long long)tn * c * h * w;
  const bool quads = levels == 8 && !count && (w & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out_val) |
                       reinterpret_cast<uintptr_t>(out_idx) | reinterpret_cast<uintptr_t>(level_tab)) & 15) == 0;
  if (quads)
    hipLaunchKernelGGL(quant8x4_kernel, dim3(pconv_grid(total / 4)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(x), level_tab, reinterpret_cast<float4 *>(out_val),
                       reinterpret_cast<float4 *>(out_idx), widths, c, h * w, w, npart, total / 4);
  else
    hipLaunchKernelGGL(quant_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream), x,
                       level_tab, out_val, out_idx, count, widths, c, h * w, w, levels, npart, total);
  PCONV_LAUNCH_CHECK("quant");
  return PCONV_OK;
}

extern "C" int pconv_dquant(const float *x, const float *weight, float *level_tab, float *out,
                            const int32_t *widths, int tn, int c, int h, int w, int wc, int levels,
                            int npart, void *stream) {
  PCONV_REQUIRE(x && weight && level_tab && out && widths, "dquant: null pointer");
  PCONV_REQUIRE(tn > 0 && c > 0 && c <= wc && h > 0 && w > 0, "dquant: bad shape c=%d wc=%d", c, wc);
  hipLaunchKernelGGL(dquant_levels_kernel, dim3((wc + 255) / 256), dim3(256), 0, as_stream(stream),
                     weight, level_tab, wc, levels);
  const long long total = (long long)tn * c * h * w;
  const bool quads = levels == 8 && (w & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                       reinterpret_cast<uintptr_t>(level_tab)) & 15) == 0;
  if (quads)
    hipLaunchKernelGGL(dquant8x4_kernel, dim3(pconv_grid(total / 4)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(x), level_tab, reinterpret_cast<float4 *>(out), widths, c, h * w,
                       w, npart, total / 4);
  else
    hipLaunchKernelGGL(dquant_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream), x,
                       level_tab, out, widths, c, h * w, w, levels, npart, total);
  PCONV_LAUNCH_CHECK("dquant");
  return PCONV_OK;
}

// ClipData.forward (model_zoo_v2.py:8-26): identity on [0, 1], slope 0.01 outside, as the reference writes it --
// x * 0.01 below 0, 1 + (x - 1) * 0.01 above 1, every operation rounded on its own (-ffp-contract=off) -- in place,
// one pass (the reference's masked index assignments are two nonzero() passes with a host synchronisation each).
__device__ __forceinline__ float leaky_clip(float v) {
  if (v < 0.f) return v * 0.01f;
  if (v > 1.f) {
    const float d = v - 1.f;
    const float s = d * 0.01f;
    return 1.f + s;
  }
  return v;
}

__global__ void leaky_clip_kernel(float *__restrict__ x, long long n, int vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    float4 *x4 = reinterpret_cast<float4 *>(x);
    const long long n4 = n / 4;
    for (long long k = i; k < n4; k += stride) {
      float4 v = x4[k];
      v.x = leaky_clip(v.x), v.y = leaky_clip(v.y), v.z = leaky_clip(v.z), v.w = leaky_clip(v.w);
      x4[k] = v;
    }
    for (long long k = n4 * 4 + i; k < n; k += stride) x[k] = leaky_clip(x[k]);
  } else {
    for (; i < n; i += stride) x[i] = leaky_clip(x[i]);
  }
}

extern "C" int pconv_leaky_clip(float *x, long long n, void *stream) {
  PCONV_REQUIRE(x && n >= 0, "leaky_clip: bad argument");
  if (n == 0) return PCONV_OK;
  const int vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  hipLaunchKernelGGL(leaky_clip_kernel, dim3(pconv_grid(vec ? (n + 3) / 4 : n)), dim3(kBlock), 0, as_stream(stream), x, n, vec);
  PCONV_LAUNCH_CHECK("leaky_clip");
  return PCONV_OK;
}

// Frame I/O at the PCIe boundary (pseudo_codec.py:215-221).  The reference converts on the host and moves fp32 over
// the bus: img2tensor = uint8 HWC -> float32 CHW / 255. -> .to(device); tensor2img = (data * 255.).to('cpu') ->
// astype(uint8).  Here the bus carries the uint8 image (a quarter of the bytes) and the SAME arithmetic runs on the
// device: float(u8) / 255.f correctly rounded (numpy's float32 division), and (uint8)(int)(v * 255.f) -- numpy's
// float32 -> uint8 cast truncates toward zero and keeps the low byte (ClipData leaks past [0, 1]: 1.004 * 255 = 256.02
// -> 0 there and here).  One thread owns 4 consecutive pixels of a row: 12 bytes of the interleaved image as three
// 32-bit words, one 16-byte piece per colour plane.
__global__ void frames_u8_to_f32_kernel(const uint32_t *__restrict__ in, float *__restrict__ out, long long plane,
                                        long long nquad) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long qpp = plane / 4;  // pixel quads per frame
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
    const uint32_t a = in[3 * q], b = in[3 * q + 1], c = in[3 * q + 2];
    // bytes: a = p0c0 p0c1 p0c2 p1c0 | b = p1c1 p1c2 p2c0 p2c1 | c = p2c2 p3c0 p3c1 p3c2
    const unsigned px[4][3] = {{a & 255u, (a >> 8) & 255u, (a >> 16) & 255u},
                               {a >> 24, b & 255u, (b >> 8) & 255u},
                               {(b >> 16) & 255u, b >> 24, c & 255u},
                               {(c >> 8) & 255u, (c >> 16) & 255u, c >> 24}};
    const long long n = q / qpp, r = q - n * qpp;
    float *base = out + n * 3 * plane + 4 * r;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      float4 v;
      v.x = __fdiv_rn((float)px[0][ch], 255.f);
      v.y = __fdiv_rn((float)px[1][ch], 255.f);
      v.z = __fdiv_rn((float)px[2][ch], 255.f);
      v.w = __fdiv_rn((float)px[3][ch], 255.f);
      *reinterpret_cast<float4 *>(base + ch * plane) = v;
    }
  }
}

__device__ __forceinline__ unsigned to_u8_like_numpy(float v) { return (unsigned)(int)(v * 255.f) & 255u; }

__global__ void frames_f32_to_u8_kernel(const float *__restrict__ in, uint32_t *__restrict__ out, long long plane,
                                        long long nquad) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long qpp = plane / 4;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
    const long long n = q / qpp, r = q - n * qpp;
    const float *base = in + n * 3 * plane + 4 * r;
    unsigned px[4][3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float4 v = *reinterpret_cast<const float4 *>(base + ch * plane);
      px[0][ch] = to_u8_like_numpy(v.x), px[1][ch] = to_u8_like_numpy(v.y);
      px[2][ch] = to_u8_like_numpy(v.z), px[3][ch] = to_u8_like_numpy(v.w);
    }
    out[3 * q] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
    out[3 * q + 1] = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
    out[3 * q + 2] = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
  }
}

extern "C" int pconv_frames_u8_to_f32(const uint8_t *in, float *out, int n, int height, int width, void *stream) {
  PCONV_REQUIRE(in && out && n > 0 && height > 0 && width > 0, "frames_u8_to_f32: bad argument");
  PCONV_REQUIRE(width % 4 == 0, "frames_u8_to_f32: the width must be a multiple of 4");
  PCONV_REQUIRE((reinterpret_cast<uintptr_t>(in) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "frames_u8_to_f32: the image must be 4-byte aligned, the tensor 16-byte aligned");
  const long long plane = (long long)height * width, nquad = (long long)n * plane / 4;
  hipLaunchKernelGGL(frames_u8_to_f32_kernel, dim3(pconv_grid(nquad)), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<const uint32_t *>(in), out, plane, nquad);
  PCONV_LAUNCH_CHECK("frames_u8_to_f32");
  return PCONV_OK;
}

extern "C" int pconv_frames_f32_to_u8(const float *in, uint8_t *out, int n, int height, int width, void *stream) {
  PCONV_REQUIRE(in && out && n > 0 && height > 0 && width > 0, "frames_f32_to_u8: bad argument");
  PCONV_REQUIRE(width % 4 == 0, "frames_f32_to_u8: the width must be a multiple of 4");
  PCONV_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0,
                "frames_f32_to_u8: the image must be 4-byte aligned, the tensor 16-byte aligned");
  const long long plane = (long long)height * width, nquad = (long long)n * plane / 4;
  hipLaunchKernelGGL(frames_f32_to_u8_kernel, dim3(pconv_grid(nquad)), dim3(kBlock), 0, as_stream(stream), in,
                     reinterpret_cast<uint32_t *>(out), plane, nquad);
  PCONV_LAUNCH_CHECK("frames_f32_to_u8");
  return PCONV_OK;
}

extern "C" int pconv_project(const float *in, const float *tf, float *out, int n, int c,
                             int height, int width, int nview, int h_out, int w_out, int nearest,
                             void *stream) {
  PCONV_REQUIRE(in && tf && out, "project: null pointer");
  PCONV_REQUIRE(n > 0 && c > 0 && nview > 0, "project: bad shape");
  const long long total = (long long)n * c * nview * h_out * w_out;
  hipLaunchKernelGGL(project_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream),
                     in, tf, out, h_out * w_out, height, width, n * c, nearest, total);
  PCONV_LAUNCH_CHECK("project");
  return PCONV_OK;
}

extern "C" int pconv_context_reshape(const float *in, float *out, int n, int c, int h, int w,
                                     int ngroup, void *stream) {
  PCONV_REQUIRE(in && out && ngroup > 0 && c % ngroup == 0, "context_reshape: bad argument");
  const long long total = (long long)n * c * h * w;
  hipLaunchKernelGGL(context_reshape_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0,
                     as_stream(stream), in, out, h * w, c, c / ngroup, total);
  PCONV_LAUNCH_CHECK("context_reshape");
  return PCONV_OK;
}

extern "C" int pconv_mask_constrain(float *weight, int nout, int cin, int k, int ngroup,
                                    int constrain, void *stream) {
  PCONV_REQUIRE(weight && ngroup > 0 && nout % ngroup == 0 && cin % ngroup == 0,
                "mask_constrain: bad argument");
  const int total = nout * cin * k * k;
  hipLaunchKernelGGL(mask_constrain_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     as_stream(stream), weight, cin, k, cin / ngroup, nout / ngroup, constrain,
                     total);
  PCONV_LAUNCH_CHECK("mask_constrain");
  return PCONV_OK;
}

extern "C" int pconv_gmm_loss(const float *weight, const float *delta, const float *mean,
                              const float *label, float *loss, float *d_weight, float *d_delta,
                              float *d_mean, float *d_label, int m, int ng, void *stream) {
  PCONV_REQUIRE(weight && delta && mean && label && loss, "gmm_loss: null pointer");
  PCONV_REQUIRE(!d_weight || (d_delta && d_mean && d_label), "gmm_loss: partial gradient buffers");
  hipLaunchKernelGGL(gmm_loss_kernel, dim3((m + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     as_stream(stream), weight, delta, mean, label, loss, d_weight, d_delta, d_mean,
                     d_label, m, ng);
  PCONV_LAUNCH_CHECK("gmm_loss");
  return PCONV_OK;
}

// PseudoQuantOp.backward: x, val, idx = the forward's input and outputs; level_tab = the table the
// forward left (w[c,0], exp(w[c,j])); g_idx may be null (ntop == 1).  bins: scratch (c*levels).
extern "C" int pconv_quant_backward(const float *x, const float *val, const float *idx, const float *g_val,
                                    const float *g_idx, const float *level_tab, float *g_in, float *g_weight,
                                    float *bins, const int32_t *widths, float top_alpha, int tn, int c, int h,
                                    int w, int levels, int npart, void *stream) {
  PCONV_REQUIRE(x && val && idx && g_val && level_tab && g_in && g_weight && bins && widths,
                "quant_backward: null pointer");
  PCONV_REQUIRE(tn > 0 && c > 0 && h > 0 && w > 0 && levels > 1, "quant_backward: bad shape");
  if (hipMemsetAsync(bins, 0, (size_t)c * levels * sizeof(float), as_stream(stream)) != hipSuccess) {
    pconv_set_error("quant_backward: memset failed");
    return PCONV_ELAUNCH;
  }
  const long long total = (long long)tn * c * h * w;
  hipLaunchKernelGGL(quant_backward_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream), x, val, idx,
                     g_val, g_idx, level_tab, g_in, bins, widths, top_alpha, c, h * w, w, levels, npart, total);
  hipLaunchKernelGGL(quant_weight_grad_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), bins,
                     level_tab, g_weight, c, levels);
  PCONV_LAUNCH_CHECK("quant_backward");
  return PCONV_OK;
}

// ProjectsOp.backward: gout (n*nview, c, h_out, w_out) -> gin, count (n, c, height, width)
extern "C" int pconv_project_backward(const float *gout, const float *tf, float *gin, float *count, int n, int c,
                                      int height, int width, int nview, int h_out, int w_out, int nearest,
                                      void *stream) {
  PCONV_REQUIRE(gout && tf && gin && count, "project_backward: null pointer");
  PCONV_REQUIRE(n > 0 && c > 0 && nview > 0, "project_backward: bad shape");
  const size_t bytes = (size_t)n * c * height * width * sizeof(float);
  if (hipMemsetAsync(gin, 0, bytes, as_stream(stream)) != hipSuccess ||
      hipMemsetAsync(count, 0, bytes, as_stream(stream)) != hipSuccess) {
    pconv_set_error("project_backward: memset failed");
    return PCONV_ELAUNCH;
  }
  const long long total = (long long)n * c * nview * h_out * w_out;
  hipLaunchKernelGGL(project_backward_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream), gout, tf,
                     gin, count, h_out * w_out, height, width, n * c, nearest, total);
  PCONV_LAUNCH_CHECK("project_backward");
  return PCONV_OK;
}
