// SphereSlice / SphereUslice: horizontal 4-tap cubic resampling between the ERP
// image and the latitude-tile stack (reference: sphere_slice_cuda.cu:87-146,
// sphere_uslice_cuda.cu:73-126).
//
// HBM-bound (read once + write once).  One workgroup stages up to RB source rows
// in LDS with coalesced loads, keeps the tap record of a column in registers and
// reuses it for all RB rows, so the tap table (20 B/column, L2 resident) costs
// 20/RB bytes per output element.  Lanes map to consecutive output columns, so
// stores are contiguous 256 B per wave.
#include <stdlib.h>
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

// value of the 4-tap filter on a circular row of `period` samples held in LDS.
// Summation order is the reference's: ((c0*a + c1*b) + c2*c) + c3*d.
__device__ __forceinline__ float cubic_at(const float *row, const Tap &t, int period) {
  int ia = t.col - 1;
  ia += (ia < 0) ? period : 0;
  int ic = t.col + 1;
  ic -= (ic >= period) ? period : 0;
  int id = t.col + 2;
  id -= (id >= period) ? period : 0;
  return t.c0 * row[ia] + t.c1 * row[t.col] + t.c2 * row[ic] + t.c3 * row[id];
}

// grid.x = n*c*height/rb row groups.  out interior (pad offset) only.
__global__ __launch_bounds__(kBlock) void slice_kernel(
    const float *__restrict__ in, float *__restrict__ out, const int32_t *__restrict__ widths,
    const int32_t *__restrict__ tap_col, const float *__restrict__ tap_coef, int c, int height,
    int width, int npart, int pad, int rb, long long ngroups, int vec4) {
  extern __shared__ float lds[];
  const int th_tile = height / npart;
  const int oh = th_tile + 2 * pad, ow = width + 2 * pad;
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    // g enumerates (image-channel, first row); rows of a group share one tile
    const long long row0 = g * rb;
    const int ph = (int)(row0 % height);
    const long long nc = row0 / height;
    const int pt = ph / th_tile;
    const int th = ph - pt * th_tile;
    const int pn = (int)(nc / c);
    const int pc = (int)(nc % c);
    const float *src = in + (size_t)row0 * width;
    __syncthreads();
    for (int i = threadIdx.x * 4; i < rb * width; i += kBlock * 4) {
      if (((width & 3) == 0)) {
        *reinterpret_cast<float4 *>(lds + i) = *reinterpret_cast<const float4 *>(src + i);
      } else {
        for (int k = 0; k < 4 && i + k < rb * width; k++) lds[i + k] = src[i + k];
      }
    }
    __syncthreads();
    const int valid = widths[pt];
    float *dst = out + (((size_t)(pn * npart + pt) * c + pc) * oh + th + pad) * ow + pad;
    if (vec4) {
      // four consecutive output columns per lane: one 16-byte store per row (the tap records of the four columns
      // are four 16-byte loads + one of their columns), a quarter of the memory instructions of the scalar form
      for (int tw = threadIdx.x * 4; tw < width; tw += kBlock * 4) {
        const size_t e = (size_t)pt * width + tw;
        const int4 cols = *reinterpret_cast<const int4 *>(tap_col + e);
        const int cc[4] = {cols.x, cols.y, cols.z, cols.w};
        Tap t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float4 cf = *reinterpret_cast<const float4 *>(tap_coef + (e + k) * 4);
          t[k].col = cc[k], t[k].c0 = cf.x, t[k].c1 = cf.y, t[k].c2 = cf.z, t[k].c3 = cf.w;
        }
        for (int r = 0; r < rb; r++) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = tw + k < valid ? cubic_at(lds + r * width, t[k], width) : 0.f;
          *reinterpret_cast<float4 *>(dst + (size_t)r * ow + tw) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else {
      for (int tw = threadIdx.x; tw < width; tw += kBlock) {
        if (tw < valid) {
          const Tap t = load_tap(tap_col, tap_coef, (size_t)pt * width + tw);
          for (int r = 0; r < rb; r++) dst[(size_t)r * ow + tw] = cubic_at(lds + r * width, t, width);
        } else {
          for (int r = 0; r < rb; r++) dst[(size_t)r * ow + tw] = 0.f;
        }
      }
    }
  }
}

// grid.x over (n*c*h*npart)/rb output row groups
__global__ __launch_bounds__(kBlock) void uslice_kernel(
    const float *__restrict__ in, float *__restrict__ out, const int32_t *__restrict__ widths,
    const int32_t *__restrict__ tap_col, const float *__restrict__ tap_coef, int c, int h,
    int width, int npart, int pad, int rb, long long ngroups, int vec4) {
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
    const float *src = in + (((size_t)(pn * npart + pb) * c + pc) * ih + ph + pad) * iw + pad;
    __syncthreads();
    for (int r = 0; r < rb; r++)
      for (int i = threadIdx.x; i < valid; i += kBlock) lds[r * width + i] = src[(size_t)r * iw + i];
    __syncthreads();
    float *dst = out + (size_t)row0 * width;
    if (vec4) {
      for (int tw = threadIdx.x * 4; tw < width; tw += kBlock * 4) {
        const size_t e = (size_t)pb * width + tw;
        const int4 cols = *reinterpret_cast<const int4 *>(tap_col + e);
        const int cc[4] = {cols.x, cols.y, cols.z, cols.w};
        Tap t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float4 cf = *reinterpret_cast<const float4 *>(tap_coef + (e + k) * 4);
          t[k].col = cc[k], t[k].c0 = cf.x, t[k].c1 = cf.y, t[k].c2 = cf.z, t[k].c3 = cf.w;
        }
        for (int r = 0; r < rb; r++) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = cubic_at(lds + r * width, t[k], valid);
          *reinterpret_cast<float4 *>(dst + (size_t)r * width + tw) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else {
      for (int tw = threadIdx.x; tw < width; tw += kBlock) {
        const Tap t = load_tap(tap_col, tap_coef, (size_t)pb * width + tw);
        for (int r = 0; r < rb; r++) dst[(size_t)r * width + tw] = cubic_at(lds + r * width, t, valid);
      }
    }
  }
}

int rows_per_block(int tile_rows, int width) {
  // PCONV_RESAMPLE_ROWS (1 / 2 / 4): rows a workgroup stages -- fewer rows = less LDS per workgroup = more of them
  // resident per CU against 20 / rows bytes of tap table per output element
  // (measured, 1x3x2048x4096: 4 rows 3.47 / 3.20 TB/s slice / uslice, 2 rows 3.92 / 3.71, 1 row 3.79 / 3.51)
  static const int cap = getenv("PCONV_RESAMPLE_ROWS") ? atoi(getenv("PCONV_RESAMPLE_ROWS")) : 2;
  int rb = cap >= 1 && cap <= kMaxRows ? cap : kMaxRows;
  while (rb > 1 && (tile_rows % rb != 0 || (size_t)rb * width * 4 > 64 * 1024)) rb >>= 1;
  return rb;
}

}  // namespace

extern "C" int pconv_sphere_slice(const float *in, float *out, const int32_t *widths,
                                  const int32_t *tap_col, const float *tap_coef, int n, int c,
                                  int height, int width, int npart, int pad, void *stream) {
  PCONV_REQUIRE(in && out && widths && tap_col && tap_coef, "sphere_slice: null pointer");
  PCONV_REQUIRE(n > 0 && c > 0 && npart > 0 && height % npart == 0 && pad >= 0,
                "sphere_slice: bad shape n=%d c=%d h=%d npart=%d", n, c, height, npart);
  PCONV_REQUIRE((size_t)width * 4 <= 64 * 1024, "sphere_slice: width %d exceeds LDS row", width);
  const int rb = rows_per_block(height / npart, width);
  const long long ngroups = (long long)n * c * height / rb;
  const unsigned grid = (unsigned)(ngroups < 256 * 16 ? ngroups : 256 * 16);
  // 16-byte stores / tap loads: whole quads per row, rows 16-byte aligned (no pad offset), aligned tensors
  const int vec4 = (width % 4 == 0) && pad == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(tap_col) |
                                                      reinterpret_cast<uintptr_t>(tap_coef)) & 15) == 0;
  hipLaunchKernelGGL(slice_kernel, dim3(grid), dim3(kBlock), (size_t)rb * width * 4,
                     as_stream(stream), in, out, widths, tap_col, tap_coef, c, height, width,
                     npart, pad, rb, ngroups, vec4);
  PCONV_LAUNCH_CHECK("sphere_slice");
  return PCONV_OK;
}

extern "C" int pconv_sphere_uslice(const float *in, float *out, const int32_t *widths,
                                   const int32_t *tap_col, const float *tap_coef, int n, int c,
                                   int h, int width, int npart, int pad, void *stream) {
  PCONV_REQUIRE(in && out && widths && tap_col && tap_coef, "sphere_uslice: null pointer");
  PCONV_REQUIRE(n > 0 && c > 0 && npart > 0 && h > 0 && pad >= 0, "sphere_uslice: bad shape");
  PCONV_REQUIRE((size_t)width * 4 <= 64 * 1024, "sphere_uslice: width %d exceeds LDS row", width);
  const int rb = rows_per_block(h, width);
  const long long ngroups = (long long)n * c * h * npart / rb;
  const unsigned grid = (unsigned)(ngroups < 256 * 16 ? ngroups : 256 * 16);
  const int vec4 = (width % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(tap_col) |
                                          reinterpret_cast<uintptr_t>(tap_coef)) & 15) == 0;
  hipLaunchKernelGGL(uslice_kernel, dim3(grid), dim3(kBlock), (size_t)rb * width * 4,
                     as_stream(stream), in, out, widths, tap_col, tap_coef, c, h, width, npart,
                     pad, rb, ngroups, vec4);
  PCONV_LAUNCH_CHECK("sphere_uslice");
  return PCONV_OK;
}
