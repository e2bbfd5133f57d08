// PseudoPad (fused), PseudoFill, Dtow/Wtod.
//
// PseudoPad: the reference runs three launches (interior copy + zero, vertical
// halo lerp from the neighbouring tile, horizontal circular wrap;
// pseudo_pad.cu:39-96).  The wrap reads what the first two wrote, so fused in one
// pass every output element is defined directly:
//     out(row r, col j) = F(r, wrap(j))        j maps into [0, valid) or is dead
//     F(r, i) = in(r - pad, i)                               interior rows
//             = in_nb(sr, q)*t + in_nb(sr, (q+1) % valid_nb)*(1-t)   halo rows
// HBM-bound: reads the tensor once, writes the padded tensor once.
#include "common.h"

namespace {

constexpr int kBlock = 256;

// one workgroup per output row at a time; row decode is wave-uniform.
__global__ __launch_bounds__(kBlock) void pseudo_pad_kernel(
    const float *__restrict__ in, float *__restrict__ out, const int32_t *__restrict__ widths,
    const int32_t *__restrict__ src_tile, const int32_t *__restrict__ src_row,
    const int32_t *__restrict__ col, const float *__restrict__ wgt, int c, int h, int w, int pad,
    int npart, long long nrows) {
  const int oh = h + 2 * pad, ow = w + 2 * pad;
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
    if (interior) {
      const float *src = in + ((size_t)tc * h + (r - pad)) * w;
      for (int j = threadIdx.x; j < ow; j += kBlock) {
        int i = j - pad;
        i += (i < 0) ? valid : 0;
        i -= (j >= valid + pad) ? valid : 0;
        dst[j] = (j < valid + 2 * pad) ? src[i] : 0.f;
      }
    } else {
      const int side = (r < pad) ? 0 : 1;
      const int rr = side ? r - pad - h : r;
      const int e = (tg * 2 + side) * pad + rr;
      const int st = src_tile[e];
      const int svalid = widths[st];
      const float *src = in + (((size_t)(img * npart + st) * c + pc) * h + src_row[e]) * w;
      const int32_t *ecol = col + (size_t)e * w;
      const float *ewgt = wgt + (size_t)e * w;
      for (int j = threadIdx.x; j < ow; j += kBlock) {
        int i = j - pad;
        i += (i < 0) ? valid : 0;
        i -= (j >= valid + pad) ? valid : 0;
        float v = 0.f;
        if (j < valid + 2 * pad) {
          const int q = ecol[i];
          const float t = ewgt[i];
          int q1 = q + 1;
          q1 = (q1 >= svalid) ? q1 - svalid : q1;
          v = src[q] * t + src[q1] * (1 - t);
        }
        dst[j] = v;
      }
    }
  }
}

// PseudoPad when the tensor already lives in the interior of a padded buffer
// (written there by the convolution that produced it): only the ring is computed.
// buf: (tn, c, h + 2*store, w + 2*store), the data at offset (store, store); the
// padded tensor of pad p <= store is the sub-view starting at (store - p, store - p).
// Per interior row 3p + (store - p) elements: left wrap, right wrap, the ring
// columns to the right of the interior (zero unless the wrap reaches them) and
// the interior columns a wider pad of the same buffer could have written before;
// per halo row the whole row, as pseudo_pad_kernel defines it.  Interior columns
// past the wrap are the producer's (it trims them to zero).
// Workgroup = (tile, halo row x 256-column strip | the side columns, group of kRingChannels channels).  A thread
// owns ONE column of a halo row: its source column pair and weight come from the tables once (they depend on the
// tile and the row only) and stay in registers while it walks the group's channels -- two gathers and one store
// per element, nothing dependent in front of them, several channels in flight.  (r3's form, one workgroup per
// (tile, channel) plane, re-read the tables for every channel -- index -> gather -> store, three dependent round
// trips per element and 25 000 tiny workgroups per launch: 1 TB/s of ring written; r3's first form decoded a flat
// 64-bit index per element: 0.6 TB/s.)  Same arithmetic per element: identical bits.
#ifndef PCONV_RING_CHANNELS
#define PCONV_RING_CHANNELS 8
#endif
constexpr int kRingChannels = PCONV_RING_CHANNELS;

__global__ __launch_bounds__(kBlock) void pseudo_pad_ring_kernel(
    float *__restrict__ buf, const int32_t *__restrict__ widths, const int32_t *__restrict__ src_tile,
    const int32_t *__restrict__ src_row, const int32_t *__restrict__ col, const float *__restrict__ wgt,
    int c, int h, int w, int pad, int store, int npart, int strips) {
  const int sh = h + 2 * store, sw = w + 2 * store;  // storage extent
  const int ow = w + 2 * pad, shift = store - pad;   // view extent / view -> storage offset
  const int per_row = 3 * pad + shift;
  const unsigned tb = blockIdx.x;                    // tile-batch index: image * npart + tile
  const int tg = (int)(tb % (unsigned)npart);
  const unsigned img = tb / (unsigned)npart;
  const int c0 = blockIdx.z * kRingChannels;
  const int nc = c - c0 < kRingChannels ? c - c0 : kRingChannels;
  const int valid = widths[tg];
  const size_t cstride = (size_t)sh * sw;            // channel stride of the storage
  float *plane0 = buf + ((size_t)tb * c + c0) * cstride;
  const int slot = blockIdx.y;
  if (slot < 2 * pad * strips) {
    // a halo row: 2 * pad rows of ow elements
    const int q = slot / strips, j = (slot % strips) * kBlock + threadIdx.x;
    if (j >= ow) return;
    const int side = q >= pad, rr = side ? q - pad : q;
    const int rview = side ? pad + h + rr : rr;
    const int e = (tg * 2 + side) * pad + rr;
    float *dst = plane0 + (size_t)(rview + shift) * sw + shift + j;
    if (j >= valid + 2 * pad) {
      for (int k = 0; k < nc; k++) dst[k * cstride] = 0.f;
      return;
    }
    const int st = src_tile[e];
    const int svalid = widths[st];
    int x = j - pad;
    x += (x < 0) ? valid : 0;
    x -= (j >= valid + pad) ? valid : 0;
    const int qc = col[(size_t)e * w + x];
    const float t = wgt[(size_t)e * w + x];
    int q1 = qc + 1;
    q1 = (q1 >= svalid) ? q1 - svalid : q1;
    const float *src = buf + ((((size_t)(img * npart + st) * c + c0) * sh) + store + src_row[e]) * sw + store;
#pragma unroll 4
    for (int k = 0; k < nc; k++) dst[k * cstride] = src[k * cstride + qc] * t + src[k * cstride + q1] * (1 - t);
    return;
  }
  // side columns of the interior rows: per row 3 * pad + shift elements, a thread per (channel, row, k)
  const int per_plane = h * per_row;
  for (int i = threadIdx.x; i < nc * per_plane; i += kBlock) {
    const int k0 = i / per_plane, i1 = i - k0 * per_plane;
    const int r = i1 / per_row, k = i1 - r * per_row;
    float *line = plane0 + k0 * cstride + (size_t)(store + r) * sw;  // storage row of data row r
    const float *data = line + store;
    int j;  // view column
    float v = 0.f;
    if (k < pad) {
      j = k;
      v = data[valid - pad + k];
    } else if (k < 2 * pad) {
      j = pad + valid + (k - pad);
      v = data[k - pad];
    } else if (k < 3 * pad) {
      j = pad + w + (k - 2 * pad);
      if (j < valid + 2 * pad) continue;  // the wrap wrote it
    } else {
      j = valid + 2 * pad + (k - 3 * pad);
      if (j >= pad + w) continue;  // outside the interior: handled as ring
    }
    line[j + shift] = v;
  }
}

__global__ __launch_bounds__(kBlock) void pseudo_fill_kernel(float *__restrict__ data,
                                                             const int32_t *__restrict__ widths,
                                                             int c, int h, int w, int npart,
                                                             int pad, int trim, float fvalue,
                                                             long long nrows) {
  for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int r = (int)(row % h);
    const int tg = (int)((row / h / c) % npart);
    float *dst = data + (size_t)row * w;
    if (r < pad - trim || r >= h - pad + trim) {
      for (int j = threadIdx.x; j < w; j += kBlock) dst[j] = fvalue;
    } else {
      const int lo = pad - trim;
      const int hi = pad + widths[tg] + trim;
      for (int j = threadIdx.x; j < lo; j += kBlock) dst[j] = fvalue;
      // start at the 64-aligned column below hi so a wave's stores stay contiguous
      for (int j = (hi & ~63) + threadIdx.x; j < w; j += kBlock)
        if (j >= hi) dst[j] = fvalue;
    }
  }
}

// depth -> width, stride 2: one thread produces 4 consecutive outputs of one row
// from 2 channels x 2 input columns (float2 loads, float4 store).
__global__ __launch_bounds__(kBlock) void dtow2_kernel(const float *__restrict__ in,
                                                       float *__restrict__ out, int c_out, int h,
                                                       int w, long long nquads) {
  const int wq = w / 2;  // quads per output row (w_out = 2w, 4 outputs per quad)
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < nquads;
       q += (long long)gridDim.x * kBlock) {
    const int xq = (int)(q % wq);
    long long rest = q / wq;
    const int oh = (int)(rest % (2 * h));
    rest /= (2 * h);  // n*c_out + pc
    const int sy = oh & 1, th = oh >> 1;
    const size_t plane = (size_t)h * w;
    const float *a = in + ((size_t)rest * 4 + sy * 2) * plane + (size_t)th * w + xq * 2;
    const float2 va = *reinterpret_cast<const float2 *>(a);
    const float2 vb = *reinterpret_cast<const float2 *>(a + plane);
    float4 o;
    o.x = va.x;
    o.y = vb.x;
    o.z = va.y;
    o.w = vb.y;
    *reinterpret_cast<float4 *>(out + ((size_t)rest * 2 * h + oh) * (2 * w) + xq * 4) = o;
  }
}

// width -> depth, stride 2: inverse of the above
__global__ __launch_bounds__(kBlock) void wtod2_kernel(const float *__restrict__ in,
                                                       float *__restrict__ out, int c_in, int h,
                                                       int w, long long nquads) {
  const int wq = w / 4;  // quads per input row
  const int ho = h / 2, wo = w / 2;
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < nquads;
       q += (long long)gridDim.x * kBlock) {
    const int xq = (int)(q % wq);
    long long rest = q / wq;
    const int ih = (int)(rest % h);
    rest /= h;  // n*c_in + tc
    const float4 v = *reinterpret_cast<const float4 *>(in + ((size_t)rest * h + ih) * w + xq * 4);
    const int sy = ih & 1, ph = ih >> 1;
    const size_t plane = (size_t)ho * wo;
    float *a = out + ((size_t)rest * 4 + sy * 2) * plane + (size_t)ph * wo + xq * 2;
    *reinterpret_cast<float2 *>(a) = make_float2(v.x, v.z);
    *reinterpret_cast<float2 *>(a + plane) = make_float2(v.y, v.w);
  }
}

// generic stride, one element per thread, output-ordered (coalesced stores)
__global__ __launch_bounds__(kBlock) void dtow_generic_kernel(const float *__restrict__ in,
                                                              float *__restrict__ out, int c, int h,
                                                              int w, int s, int d2w,
                                                              long long total) {
  const int s2 = s * s;
  for (long long o = (long long)blockIdx.x * kBlock + threadIdx.x; o < total;
       o += (long long)gridDim.x * kBlock) {
    if (d2w) {
      const int wo = w * s, ho = h * s, co = c / s2;
      const int ow = (int)(o % wo);
      long long rest = o / wo;
      const int oh = (int)(rest % ho);
      rest /= ho;
      const int pc = (int)(rest % co);
      const long long tn = rest / co;
      const int rc = (oh % s) * s + (ow % s);
      out[o] = in[(((size_t)tn * c + pc * s2 + rc) * h + oh / s) * w + ow / s];
    } else {
      const int wo = w / s, ho = h / s, co = c * s2;
      const int ow = (int)(o % wo);
      long long rest = o / wo;
      const int oh = (int)(rest % ho);
      rest /= ho;
      const int pc = (int)(rest % co);
      const long long tn = rest / co;
      const int tc = pc / s2, rc = pc % s2;
      out[o] = in[(((size_t)tn * c + tc) * h + oh * s + rc / s) * w + ow * s + rc % s];
    }
  }
}

}  // namespace

extern "C" int pconv_pseudo_pad(const float *in, float *out, const int32_t *widths,
                                const int32_t *src_tile, const int32_t *src_row,
                                const int32_t *col, const float *wgt, int tn, int c, int h, int w,
                                int pad, int npart, void *stream) {
  PCONV_REQUIRE(in && out && widths, "pseudo_pad: null pointer");
  PCONV_REQUIRE(pad == 0 || (src_tile && src_row && col && wgt), "pseudo_pad: null table");
  PCONV_REQUIRE(tn > 0 && tn % npart == 0 && c > 0 && h > 0 && w > 0 && pad >= 0,
                "pseudo_pad: bad shape tn=%d c=%d h=%d w=%d pad=%d", tn, c, h, w, pad);
  const long long nrows = (long long)tn * c * (h + 2 * pad);
  const unsigned grid = (unsigned)(nrows < 256 * 32 ? nrows : 256 * 32);
  hipLaunchKernelGGL(pseudo_pad_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), in, out,
                     widths, src_tile, src_row, col, wgt, c, h, w, pad, npart, nrows);
  PCONV_LAUNCH_CHECK("pseudo_pad");
  return PCONV_OK;
}

extern "C" int pconv_pseudo_pad_ring(float *buf, const int32_t *widths, const int32_t *src_tile,
                                     const int32_t *src_row, const int32_t *col, const float *wgt, int tn,
                                     int c, int h, int w, int pad, int store, int npart, void *stream) {
  PCONV_REQUIRE(buf && widths && src_tile && src_row && col && wgt, "pseudo_pad_ring: null pointer");
  PCONV_REQUIRE(tn > 0 && tn % npart == 0 && c > 0 && h > 0 && w > 0 && pad > 0 && pad <= store,
                "pseudo_pad_ring: bad shape tn=%d c=%d h=%d w=%d pad=%d store=%d", tn, c, h, w, pad, store);
  const long long planes = (long long)tn * c;
  PCONV_REQUIRE(planes <= 0x7fffffffLL, "pseudo_pad_ring: too many planes");
  const int strips = (w + 2 * pad + kBlock - 1) / kBlock;  // 256-column strips of a halo row
  const int cgroups = (c + kRingChannels - 1) / kRingChannels;
  PCONV_REQUIRE(2 * pad * strips + 1 <= 65535 && cgroups <= 65535, "pseudo_pad_ring: grid out of range");
  hipLaunchKernelGGL(pseudo_pad_ring_kernel, dim3((unsigned)tn, (unsigned)(2 * pad * strips + 1), (unsigned)cgroups), dim3(kBlock),
                     0, as_stream(stream), buf, widths, src_tile, src_row, col, wgt, c, h, w, pad, store, npart, strips);
  PCONV_LAUNCH_CHECK("pseudo_pad_ring");
  return PCONV_OK;
}

extern "C" int pconv_pseudo_fill(float *data, const int32_t *widths, int tn, int c, int h, int w,
                                 int npart, int pad, int trim, float fvalue, void *stream) {
  PCONV_REQUIRE(data && widths, "pseudo_fill: null pointer");
  PCONV_REQUIRE(tn > 0 && c > 0 && h > 0 && w > 0, "pseudo_fill: bad shape");
  const long long nrows = (long long)tn * c * h;
  const unsigned grid = (unsigned)(nrows < 256 * 32 ? nrows : 256 * 32);
  hipLaunchKernelGGL(pseudo_fill_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), data,
                     widths, c, h, w, npart, pad, trim, fvalue, nrows);
  PCONV_LAUNCH_CHECK("pseudo_fill");
  return PCONV_OK;
}

extern "C" int pconv_dtow(const float *in, float *out, int n, int c, int h, int w, int stride,
                          int d2w, void *stream) {
  PCONV_REQUIRE(in && out, "dtow: null pointer");
  PCONV_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && stride > 0, "dtow: bad shape");
  const int s2 = stride * stride;
  if (d2w)
    PCONV_REQUIRE(c % s2 == 0, "dtow: channels %d not divisible by %d", c, s2);
  else
    PCONV_REQUIRE(h % stride == 0 && w % stride == 0, "wtod: size not divisible by stride");
  const long long total = (long long)n * c * h * w;
  if (stride == 2 && d2w && (w % 2 == 0)) {
    const long long nquads = total / 4;
    hipLaunchKernelGGL(dtow2_kernel, dim3(pconv_grid(nquads)), dim3(kBlock), 0, as_stream(stream),
                       in, out, c / 4, h, w, nquads);
  } else if (stride == 2 && !d2w && (w % 4 == 0)) {
    const long long nquads = total / 4;
    hipLaunchKernelGGL(wtod2_kernel, dim3(pconv_grid(nquads)), dim3(kBlock), 0, as_stream(stream),
                       in, out, c, h, w, nquads);
  } else {
    hipLaunchKernelGGL(dtow_generic_kernel, dim3(pconv_grid(total)), dim3(kBlock), 0,
                       as_stream(stream), in, out, c, h, w, stride, d2w, total);
  }
  PCONV_LAUNCH_CHECK("dtow");
  return PCONV_OK;
}
