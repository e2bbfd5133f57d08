// Backward (transposed) forms of the linear geometry ops: the start of the training path
// (SURVEY 8f-4).  The reference builds inverse lists with float atomics and gathers through
// them (sphere_uslice_cuda.cu:128-176, pseudo_context_cuda.cu:106-138, pseudo_pad.cu:127-175)
// or scatters with global atomics (sphere_slice_cuda.cu:191-223): the order of its sums is
// not defined.  Here every op is the exact transpose of the forward kernel of this library:
//   * slice / uslice: one workgroup per group of rows accumulates the scattered taps of a
//     row in LDS (ds_add_f32) and writes the row once, coalesced;
//   * pad: a gather -- interior + folded wrap columns + the halo entries that read the
//     element (reverse CSR built on the host, pconv_host_pad_reverse): fixed order;
//   * context reshape: the inverse permutation.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxRows = 4;

struct Tap {
  int col;
  float c0, c1, c2, c3;
};

__device__ __forceinline__ Tap load_tap(const int32_t *tap_col, const float *tap_coef, size_t e) {
  Tap t;
  t.col = tap_col[e];
  const float4 c = *reinterpret_cast<const float4 *>(tap_coef + e * 4);
  t.c0 = c.x;
  t.c1 = c.y;
  t.c2 = c.z;
  t.c3 = c.w;
  return t;
}

// transpose of cubic_at (resample.hip): the 4 taps of output column t scatter v into a
// circular row of `period` accumulators in LDS
__device__ __forceinline__ void cubic_scatter(float *row, const Tap &t, int period, float v) {
  int ia = t.col - 1;
  ia += (ia < 0) ? period : 0;
  int ic = t.col + 1;
  ic -= (ic >= period) ? period : 0;
  int id = t.col + 2;
  id -= (id >= period) ? period : 0;
  atomicAdd(row + ia, t.c0 * v);
  atomicAdd(row + t.col, t.c1 * v);
  atomicAdd(row + ic, t.c2 * v);
  atomicAdd(row + id, t.c3 * v);
}

// grad of the tile stack (n*npart, c, h/npart + 2 pad, w + 2 pad) -> grad of the ERP image
// (sphere_slice_cuda.cu:191-244)
__global__ __launch_bounds__(kBlock) void slice_backward_kernel(
    const float *__restrict__ gout, float *__restrict__ gin, const int32_t *__restrict__ widths,
    const int32_t *__restrict__ tap_col, const float *__restrict__ tap_coef, int c, int height, int width,
    int npart, int pad, int rb, long long ngroups) {
  extern __shared__ float lds[];
  const int th_tile = height / npart;
  const int oh = th_tile + 2 * pad, ow = width + 2 * pad;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long long row0 = g * rb;
    const int ph = (int)(row0 % height);
    const long long nc = row0 / height;
    const int pt = ph / th_tile;
    const int th = ph - pt * th_tile;
    const int pn = (int)(nc / c);
    const int pc = (int)(nc % c);
    __syncthreads();
    for (int i = threadIdx.x; i < rb * width; i += kBlock) lds[i] = 0.f;
    __syncthreads();
    const int valid = widths[pt];
    const float *src = gout + (((size_t)(pn * npart + pt) * c + pc) * oh + th + pad) * ow + pad;
    for (int tw = threadIdx.x; tw < valid; tw += kBlock) {
      const Tap t = load_tap(tap_col, tap_coef, (size_t)pt * width + tw);
      for (int r = 0; r < rb; r++) cubic_scatter(lds + r * width, t, width, src[(size_t)r * ow + tw]);
    }
    __syncthreads();
    float *dst = gin + (size_t)row0 * width;
    for (int i = threadIdx.x; i < rb * width; i += kBlock) dst[i] = lds[i];
  }
}

// grad of the ERP image (n, c, h*npart, w) -> grad of the tile stack, zero outside the valid
// interior (sphere_uslice_cuda.cu:128-200); gin must be zeroed by the caller
__global__ __launch_bounds__(kBlock) void uslice_backward_kernel(
    const float *__restrict__ gout, float *__restrict__ gin, const int32_t *__restrict__ widths,
    const int32_t *__restrict__ tap_col, const float *__restrict__ tap_coef, int c, int h, int width, int npart,
    int pad, int rb, long long ngroups) {
  extern __shared__ float lds[];
  const int h_out = h * npart;
  const int ih = h + 2 * pad, iw = width + 2 * pad;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long long row0 = g * rb;
    const int th = (int)(row0 % h_out);
    const long long nc = row0 / h_out;
    const int pb = th / h;
    const int ph = th - pb * h;
    const int pn = (int)(nc / c);
    const int pc = (int)(nc % c);
    const int valid = widths[pb];
    __syncthreads();
    for (int i = threadIdx.x; i < rb * width; i += kBlock) lds[i] = 0.f;
    __syncthreads();
    const float *src = gout + (size_t)row0 * width;
    for (int tw = threadIdx.x; tw < width; tw += kBlock) {
      const Tap t = load_tap(tap_col, tap_coef, (size_t)pb * width + tw);
      for (int r = 0; r < rb; r++) cubic_scatter(lds + r * width, t, valid, src[(size_t)r * width + tw]);
    }
    __syncthreads();
    float *dst = gin + (((size_t)(pn * npart + pb) * c + pc) * ih + ph + pad) * iw + pad;
    for (int r = 0; r < rb; r++)
      for (int i = threadIdx.x; i < valid; i += kBlock) dst[(size_t)r * iw + i] = lds[r * width + i];
  }
}

int rows_per_block(int tile_rows, int width) {
  int rb = kMaxRows;
  while (rb > 1 && (tile_rows % rb != 0 || (size_t)rb * width * 4 > 64 * 1024)) rb >>= 1;
  return rb;
}

// gradient that reaches padded element (row pointer, unpadded column col) of a tile whose
// valid width is wl, after the circular wrap columns have been folded back onto the interior
// columns they were copied from (transpose of pseudo_pad.cu:82-96)
template <bool CAUSAL>
__device__ __forceinline__ float folded(const float *rowp, int col, int wl, int pad) {
  float v = rowp[col + pad];
  if (col < pad) v += rowp[wl + pad + col];                     // right halo column that copied this one
  if (!CAUSAL && col >= wl - pad) v += rowp[col - (wl - pad)];  // left halo column that copied this one
  return v;  // (the causal pad of the entropy model leaves the left halo columns zero)
}

// one thread per input element (tn, c, row, col)
template <bool CAUSAL>
__global__ __launch_bounds__(kBlock) void pad_backward_kernel(
    const float *__restrict__ gout, float *__restrict__ gin, const int32_t *__restrict__ widths,
    const int32_t *__restrict__ rev_start, const int32_t *__restrict__ rev_dst, const float *__restrict__ rev_wgt,
    int c, int h, int w, int npart, int pad, long long total) {
  const int hp = h + 2 * pad, wp = w + 2 * pad;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int col = (int)(i % w);
    const int row = (int)((i / w) % h);
    const long long plane = i / ((long long)h * w);  // tn*c + pc
    const int pc = (int)(plane % c);
    const long long tn = plane / c;
    const int t = (int)(tn % npart);
    const int wl = widths[t];
    float v = 0.f;
    if (col < wl) {
      v = folded<CAUSAL>(gout + ((size_t)plane * hp + row + pad) * wp, col, wl, pad);
      const int key = (t * h + row) * w + col;
      for (int k = rev_start[key]; k < rev_start[key + 1]; k++) {
        const int d = rev_dst[k];
        const int td = d >> 24, off = d & 0xffffff;  // destination tile, padded row*wp + unpadded column
        const int prow = off / w, dcol = off - prow * w;
        const float *rowp = gout + (((size_t)(tn - t + td) * c + pc) * hp + prow) * wp;
        v += rev_wgt[k] * folded<CAUSAL>(rowp, dcol, widths[td], pad);
      }
    }
    gin[i] = v;
  }
}

// PseudoEntropyPad.forward (pseudo_entropy_pad_cuda.cu:39-134, three launches there): the causal
// pad of the training-time entropy net.  Interior copy; halo rows lerped from the neighbouring
// tile through the causal table (pconv_host_causal_table: column -2 = no source, -1 = only the
// second tap) -- no pole mirroring; the first `pad` valid columns re-appear after the last one,
// the left halo columns stay zero.  One workgroup per output row at a time.
__global__ __launch_bounds__(kBlock) void entropy_pad_kernel(
    const float *__restrict__ in, float *__restrict__ out, const int32_t *__restrict__ widths,
    const int32_t *__restrict__ col, const float *__restrict__ wgt, int c, int h, int w, int pad, int npart,
    long long nrows) {
  const int oh = h + 2 * pad, ow = w + 2 * pad;
  const int rows_all = h * npart;
  for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int r = (int)(row % oh);
    const long long tc = row / oh;  // tile-batch * c + channel
    const int pc = (int)(tc % c);
    const long long tb = tc / c;
    const int tg = (int)(tb % npart);
    const long long img = tb / npart;
    const int valid = widths[tg];
    float *dst = out + (size_t)row * ow;
    const bool interior = (r >= pad) && (r < pad + h);
    const float *src = nullptr;
    const int32_t *ecol = nullptr;
    const float *ewgt = nullptr;
    int svalid = 1;
    if (interior) {
      src = in + ((size_t)tc * h + (r - pad)) * w;
    } else {
      const int side = (r < pad) ? 0 : 1;
      const int rr = side ? r - pad - h : r;
      const int srow = side ? (tg + 1) * h + rr : tg * h - pad + rr;
      if (srow >= 0 && srow < rows_all) {
        const int st = srow / h;
        svalid = widths[st];
        src = in + (((size_t)(img * npart + st) * c + pc) * h + (srow - st * h)) * w;
        const int e = (tg * 2 + side) * pad + rr;
        ecol = col + (size_t)e * w;
        ewgt = wgt + (size_t)e * w;
      }
    }
    for (int j = threadIdx.x; j < ow; j += kBlock) {
      int i = j - pad;
      if (i >= valid && i < valid + pad) i -= valid;  // right wrap
      float v = 0.f;
      if (src && i >= 0 && i < valid && j < valid + 2 * pad) {
        if (interior) {
          v = src[i];
        } else {
          const int q = ecol[i];
          if (q != -2) {
            const float t = ewgt[i];
            int q1 = q + 1;
            q1 = (q1 >= svalid) ? q1 - svalid : q1;
            const float a = q >= 0 ? src[q] : 0.f;
            v = a * t + src[q1] * (1 - t);
          }
        }
      }
      dst[j] = v;
    }
  }
}

// context_reshape_cuda.cu:63-72
__global__ __launch_bounds__(kBlock) void context_reshape_backward_kernel(
    const float *__restrict__ top, float *__restrict__ bottom, int inner, int channel, int cpg, long long total) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const long long pn = i / inner / channel;
    const int pc = (int)((i / inner) % channel);
    const int ps = (int)(i % inner);
    const long long tidx = (pn * inner * channel / cpg + (long long)(pc / cpg) * inner + ps) * cpg + pc % cpg;
    bottom[i] = top[tidx];
  }
}

}  // namespace

extern "C" int pconv_sphere_slice_backward(const float *gout, float *gin, const int32_t *widths,
                                           const int32_t *tap_col, const float *tap_coef, int n, int c, int height,
                                           int width, int npart, int pad, void *stream) {
  PCONV_REQUIRE(gout && gin && widths && tap_col && tap_coef, "sphere_slice_backward: null pointer");
  PCONV_REQUIRE(n > 0 && c > 0 && npart > 0 && height % npart == 0 && pad >= 0, "sphere_slice_backward: bad shape");
  PCONV_REQUIRE((size_t)width * 4 <= 64 * 1024, "sphere_slice_backward: width %d exceeds LDS row", width);
  const int rb = rows_per_block(height / npart, width);
  const long long ngroups = (long long)n * c * height / rb;
  const unsigned grid = (unsigned)(ngroups < 256 * 16 ? ngroups : 256 * 16);
  hipLaunchKernelGGL(slice_backward_kernel, dim3(grid), dim3(kBlock), (size_t)rb * width * 4, as_stream(stream), gout,
                     gin, widths, tap_col, tap_coef, c, height, width, npart, pad, rb, ngroups);
  PCONV_LAUNCH_CHECK("sphere_slice_backward");
  return PCONV_OK;
}

extern "C" int pconv_sphere_uslice_backward(const float *gout, float *gin, const int32_t *widths,
                                            const int32_t *tap_col, const float *tap_coef, int n, int c, int h,
                                            int width, int npart, int pad, void *stream) {
  PCONV_REQUIRE(gout && gin && widths && tap_col && tap_coef, "sphere_uslice_backward: null pointer");
  PCONV_REQUIRE(n > 0 && c > 0 && npart > 0 && h > 0 && pad >= 0, "sphere_uslice_backward: bad shape");
  PCONV_REQUIRE((size_t)width * 4 <= 64 * 1024, "sphere_uslice_backward: width %d exceeds LDS row", width);
  const size_t bytes = (size_t)n * npart * c * (h + 2 * pad) * (width + 2 * pad) * sizeof(float);
  if (hipMemsetAsync(gin, 0, bytes, as_stream(stream)) != hipSuccess) {
    pconv_set_error("sphere_uslice_backward: memset failed");
    return PCONV_ELAUNCH;
  }
  const int rb = rows_per_block(h, width);
  const long long ngroups = (long long)n * c * h * npart / rb;
  const unsigned grid = (unsigned)(ngroups < 256 * 16 ? ngroups : 256 * 16);
  hipLaunchKernelGGL(uslice_backward_kernel, dim3(grid), dim3(kBlock), (size_t)rb * width * 4, as_stream(stream),
                     gout, gin, widths, tap_col, tap_coef, c, h, width, npart, pad, rb, ngroups);
  PCONV_LAUNCH_CHECK("sphere_uslice_backward");
  return PCONV_OK;
}

extern "C" int pconv_pseudo_pad_backward(const float *gout, float *gin, const int32_t *widths,
                                         const int32_t *rev_start, const int32_t *rev_dst, const float *rev_wgt,
                                         int tn, int c, int h, int w, int pad, int npart, void *stream) {
  PCONV_REQUIRE(gout && gin && widths && rev_start && rev_dst && rev_wgt, "pseudo_pad_backward: null pointer");
  PCONV_REQUIRE(tn > 0 && tn % npart == 0 && c > 0 && h > 0 && w > 0 && pad > 0, "pseudo_pad_backward: bad shape");
  PCONV_REQUIRE((long long)(h + 2 * pad) * w < (1 << 24) && npart <= 128, "pseudo_pad_backward: tile too large for the packed table");
  const long long total = (long long)tn * c * h * w;
  hipLaunchKernelGGL(pad_backward_kernel<false>, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream), gout,
                     gin, widths, rev_start, rev_dst, rev_wgt, c, h, w, npart, pad, total);
  PCONV_LAUNCH_CHECK("pseudo_pad_backward");
  return PCONV_OK;
}

extern "C" int pconv_context_reshape_backward(const float *top, float *bottom, int n, int c, int h, int w,
                                              int ngroup, void *stream) {
  PCONV_REQUIRE(top && bottom && ngroup > 0 && c % ngroup == 0, "context_reshape_backward: bad argument");
  const long long total = (long long)n * c * h * w;
  hipLaunchKernelGGL(context_reshape_backward_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream),
                     top, bottom, h * w, c, c / ngroup, total);
  PCONV_LAUNCH_CHECK("context_reshape_backward");
  return PCONV_OK;
}

extern "C" int pconv_entropy_pad(const float *in, float *out, const int32_t *widths, const int32_t *col,
                                 const float *wgt, int tn, int c, int h, int w, int pad, int npart, void *stream) {
  PCONV_REQUIRE(in && out && widths && col && wgt, "entropy_pad: null pointer");
  PCONV_REQUIRE(tn > 0 && tn % npart == 0 && c > 0 && h > 0 && w > 0 && pad > 0, "entropy_pad: bad shape");
  const long long nrows = (long long)tn * c * (h + 2 * pad);
  const unsigned grid = (unsigned)(nrows < 256 * 32 ? nrows : 256 * 32);
  hipLaunchKernelGGL(entropy_pad_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), in, out, widths, col, wgt, c,
                     h, w, pad, npart, nrows);
  PCONV_LAUNCH_CHECK("entropy_pad");
  return PCONV_OK;
}

extern "C" int pconv_entropy_pad_backward(const float *gout, float *gin, const int32_t *widths,
                                          const int32_t *rev_start, const int32_t *rev_dst, const float *rev_wgt,
                                          int tn, int c, int h, int w, int pad, int npart, void *stream) {
  PCONV_REQUIRE(gout && gin && widths && rev_start && rev_dst && rev_wgt, "entropy_pad_backward: null pointer");
  PCONV_REQUIRE(tn > 0 && tn % npart == 0 && c > 0 && h > 0 && w > 0 && pad > 0, "entropy_pad_backward: bad shape");
  PCONV_REQUIRE((long long)(h + 2 * pad) * w < (1 << 24) && npart <= 128, "entropy_pad_backward: tile too large for the packed table");
  const long long total = (long long)tn * c * h * w;
  hipLaunchKernelGGL(pad_backward_kernel<true>, dim3(pconv_grid(total)), dim3(kBlock), 0, as_stream(stream), gout,
                     gin, widths, rev_start, rev_dst, rev_wgt, c, h, w, npart, pad, total);
  PCONV_LAUNCH_CHECK("entropy_pad_backward");
  return PCONV_OK;
}
