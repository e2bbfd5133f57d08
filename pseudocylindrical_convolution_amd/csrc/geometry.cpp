// Host-side geometry of the latitude-tile decomposition: tile widths, resampling
// tap tables, halo gather tables and the entropy wavefront schedule.
//
// Everything here is integer / IEEE arithmetic evaluated once per shape on the
// host and uploaded by the caller.  The arithmetic order follows the reference's
// one-off table kernels so that indices are bit-exact with them:
//   tile widths      math_cuda.cu:177-253
//   slice taps       sphere_slice_cuda.cu:13-32
//   uslice taps      sphere_uslice_cuda.cu:13-30
//   pad halo table   pseudo_context_cuda.cu:51-104
//   wavefront        entropy_context_cuda.cu:13-45
//   causal halo      entropy_context_cuda.cu:64-165,187-204
//   viewport table   projects_cuda.cu:7-165
// Unlike the reference, element offsets are never stored in fp32.
//
// Build note: compiled with -ffp-contract=off so the float expressions below are
// evaluated exactly as written (no fused multiply-add).
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void pconv_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *pconv_last_error(void) { return g_err; }
extern "C" int pconv_abi_version(void) { return 1; }
extern "C" int pconv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int pconv_host_tile_widths(const float *weight, int npart, int height, int width,
                                      int32_t *widths) {
  PCONV_REQUIRE(weight && widths && npart > 0 && width > 0, "tile_widths: bad argument");
  PCONV_REQUIRE(height % npart == 0, "tile_widths: height %d is not a multiple of npart %d",
                height, npart);
  float total = 0;
  for (int i = 0; i < npart; i++) total += weight[i];
  if (total > 3 * npart) {
    // weights given in 1/64 units of the full width (base.py:set_weight)
    for (int i = 0; i < npart; i++) {
      float scaled = weight[i] / 64 * width;
      widths[i] = static_cast<int>(scaled + 0.5);
    }
    return PCONV_OK;
  }
  // fractional weights: cosine-latitude rule
  const int rows = height / npart;
  const float pi = acos(-1.0);
  const int half = npart / 2;
  const bool even = (npart % 2 == 0);
  for (int i = 0; i < npart; i++) {
    bool centre = even ? (i == half - 1 || i == half) : (i == half);
    if (centre) {
      widths[i] = width;
      continue;
    }
    double edge = (i < half) ? (rows * (i + 1) - 0.5) : (rows * i + 0.5);
    float ww = weight[i] * width;
    widths[i] = static_cast<int>(ww * cos((edge / height - 0.5) * pi) + 0.5);
  }
  return PCONV_OK;
}

// Catmull-Rom coefficients for fractional offset t, in the reference's float
// evaluation order (sphere_slice_cuda.cu:24-30).
static inline void cubic_coef(float t, float *c) {
  float t2 = t * t;
  float t3 = t * t2;
  c[0] = (-t + 2 * t2 - t3) / 2;
  c[1] = (2 - 5 * t2 + 3 * t3) / 2;
  c[2] = (t + 4 * t2 - 3 * t3) / 2;
  c[3] = (-t2 + t3) / 2;
}

extern "C" int pconv_host_slice_taps(const int32_t *widths, int npart, int width,
                                     int32_t *tap_col, float *tap_coef) {
  PCONV_REQUIRE(widths && tap_col && tap_coef, "slice_taps: null pointer");
  memset(tap_col, 0, sizeof(int32_t) * (size_t)npart * width);
  memset(tap_coef, 0, sizeof(float) * (size_t)npart * width * 4);
  for (int t = 0; t < npart; t++) {
    const int tw = widths[t];
    for (int i = 0; i < tw && i < width; i++) {
      float pos = (i + 0.5) / tw * width - 0.5 + 1e-9;
      if (pos < 0) pos = pos + width;
      float whole = static_cast<float>(static_cast<int>(pos));
      size_t e = (size_t)t * width + i;
      tap_col[e] = static_cast<int>(whole);
      cubic_coef(pos - whole, tap_coef + e * 4);
    }
  }
  return PCONV_OK;
}

extern "C" int pconv_host_uslice_taps(const int32_t *widths, int npart, int width,
                                      int32_t *tap_col, float *tap_coef) {
  PCONV_REQUIRE(widths && tap_col && tap_coef, "uslice_taps: null pointer");
  for (int t = 0; t < npart; t++) {
    const int tw = widths[t];
    for (int i = 0; i < width; i++) {
      float pos = (i + 0.5) / width * tw - 0.5 + 1e-9;
      if (pos < 0) pos = pos + tw;
      float whole = static_cast<float>(static_cast<int>(pos));
      size_t e = (size_t)t * width + i;
      tap_col[e] = static_cast<int>(whole);
      cubic_coef(pos - whole, tap_coef + e * 4);
    }
  }
  return PCONV_OK;
}

// Column of tile `to` (valid width wto) that faces column i of tile `from`
// (valid width wfrom), as the reference computes it in fp32 from a double
// expression.  `shifted` = half-turn shift used when mirroring over a pole.
static inline float facing_column(int i, int wfrom, int wto, bool shifted) {
  float p;
  if (shifted) {
    float s = i + wfrom / 2.;
    if (s >= wfrom) s = s - wfrom;
    p = (s + 0.5) / wfrom * wto - 0.5 + 1e-9;
  } else {
    p = (i + 0.5) / wfrom * wto - 0.5 + 1e-9;
  }
  return p;
}

extern "C" int pconv_host_pad_table(const int32_t *widths, int npart, int height, int width,
                                    int pad, int32_t *src_tile, int32_t *src_row,
                                    int32_t *col, float *wgt) {
  PCONV_REQUIRE(widths && src_tile && src_row && col && wgt, "pad_table: null pointer");
  PCONV_REQUIRE(pad >= 0 && height > 0, "pad_table: bad pad/height");
  const int total_rows = height * npart;
  for (int t = 0; t < npart; t++) {
    for (int side = 0; side < 2; side++) {
      for (int r = 0; r < pad; r++) {
        int row = (side == 0) ? t * height - pad + r : (t + 1) * height + r;
        bool mirrored = false;
        if (row < 0) {
          row = -row - 1;
          mirrored = true;
        } else if (row >= total_rows) {
          row = 2 * total_rows - row - 1;
          mirrored = true;
        }
        const int st = row / height;
        const int e = (t * 2 + side) * pad + r;
        src_tile[e] = st;
        src_row[e] = row % height;
        for (int i = 0; i < width; i++) {
          size_t k = (size_t)e * width + i;
          if (i >= widths[t]) {
            col[k] = 0;
            wgt[k] = 0;
            continue;
          }
          float p = facing_column(i, widths[t], widths[st], mirrored);
          if (p < 0) p = p + widths[st];
          int whole = static_cast<int>(p);
          col[k] = whole;
          wgt[k] = whole + 1 - p;
        }
      }
    }
  }
  return PCONV_OK;
}

// Reverse of pconv_host_pad_table for PseudoPad's backward: CSR over the interior elements
// (tile*height + row)*width + col of ONE channel plane set -> the halo entries that read the
// element, as (destination tile << 24 | padded row*width + column, weight).  rev_start has
// npart*height*width + 1 entries; rev_dst / rev_wgt hold at most 2 * npart*2*pad*width.
// Returns the number of records, or a negative status.
extern "C" int pconv_host_pad_reverse(const int32_t *widths, int npart, int height, int width, int pad,
                                      int32_t *rev_start, int32_t *rev_dst, float *rev_wgt) {
  PCONV_REQUIRE(widths && rev_start && rev_dst && rev_wgt, "pad_reverse: null pointer");
  PCONV_REQUIRE(pad > 0 && height > 0 && (long long)(height + 2 * pad) * width < (1 << 24) && npart <= 128,
                "pad_reverse: bad pad/height");
  const size_t n = (size_t)npart * 2 * pad * width;
  std::vector<int32_t> st(npart * 2 * pad), sr(npart * 2 * pad), col(n);
  std::vector<float> wgt(n);
  int rc = pconv_host_pad_table(widths, npart, height, width, pad, st.data(), sr.data(), col.data(), wgt.data());
  if (rc < 0) return rc;
  const size_t keys = (size_t)npart * height * width;
  std::vector<std::vector<std::pair<int32_t, float>>> lists(keys);
  for (int t = 0; t < npart; t++)
    for (int side = 0; side < 2; side++)
      for (int r = 0; r < pad; r++) {
        const int e = (t * 2 + side) * pad + r;
        const int prow = side ? height + pad + r : r;  // padded row of the halo entry
        const int s_t = st[e], s_r = sr[e], ws = widths[s_t];
        for (int i = 0; i < widths[t]; i++) {
          const size_t k = (size_t)e * width + i;
          const int c0 = col[k], c1 = (c0 + 1) % ws;
          const int32_t dst = (t << 24) | (prow * width + i);
          lists[((size_t)s_t * height + s_r) * width + c0].push_back({dst, wgt[k]});
          lists[((size_t)s_t * height + s_r) * width + c1].push_back({dst, 1.f - wgt[k]});
        }
      }
  int total = 0;
  for (size_t k = 0; k < keys; k++) {
    rev_start[k] = total;
    for (auto &p : lists[k]) {
      rev_dst[total] = p.first;
      rev_wgt[total] = p.second;
      total++;
    }
  }
  rev_start[keys] = total;
  return total;
}

// Reverse of pconv_host_causal_table for PseudoEntropyPad's backward (same record format as
// pconv_host_pad_reverse).  Taps of weight exactly 0 are left out, as in the reference's lists
// (pseudo_entropy_context_cuda.cu:171-209).
extern "C" int pconv_host_entropy_pad_table(const int32_t *widths, int npart, int height, int width, int pad,
                                            int version, int32_t *col, float *wgt);

extern "C" int pconv_host_causal_reverse(const int32_t *widths, int npart, int height, int width, int pad,
                                         int version, int32_t *rev_start, int32_t *rev_dst, float *rev_wgt) {
  PCONV_REQUIRE(widths && rev_start && rev_dst && rev_wgt, "causal_reverse: null pointer");
  PCONV_REQUIRE(pad > 0 && height > 0 && (long long)(height + 2 * pad) * width < (1 << 24) && npart <= 128,
                "causal_reverse: bad pad/height");
  const size_t n = (size_t)npart * 2 * pad * width;
  std::vector<int32_t> col(n);
  std::vector<float> wgt(n);
  int rc = pconv_host_entropy_pad_table(widths, npart, height, width, pad, version, col.data(), wgt.data());
  if (rc < 0) return rc;
  const int rows = height * npart;
  const size_t keys = (size_t)rows * width;
  std::vector<std::vector<std::pair<int32_t, float>>> lists(keys);
  for (int t = 0; t < npart; t++)
    for (int side = 0; side < 2; side++)
      for (int r = 0; r < pad; r++) {
        const int srow = side ? (t + 1) * height + r : t * height - pad + r;
        if (srow < 0 || srow >= rows) continue;
        const int prow = side ? height + pad + r : r;
        const int ws = widths[srow / height];
        const size_t base = ((size_t)(t * 2 + side) * pad + r) * width;
        for (int i = 0; i < widths[t]; i++) {
          const int c0 = col[base + i];
          if (c0 == -2) continue;
          const float tw = wgt[base + i];
          const int32_t dst = (t << 24) | (prow * width + i);
          if (c0 >= 0 && tw > 0) lists[(size_t)srow * width + c0].push_back({dst, tw});
          if (tw < 1) lists[(size_t)srow * width + (c0 + 1) % ws].push_back({dst, 1.f - tw});
        }
      }
  int total = 0;
  for (size_t k = 0; k < keys; k++) {
    rev_start[k] = total;
    for (auto &p : lists[k]) {
      rev_dst[total] = p.first;
      rev_wgt[total] = p.second;
      total++;
    }
  }
  rev_start[keys] = total;
  return total;
}

extern "C" int pconv_host_wavefront(const int32_t *widths, int npart, int height, int width,
                                    int32_t *order, int32_t *plane_start) {
  PCONV_REQUIRE(widths && order && plane_start, "wavefront: null pointer");
  const int rows = height * npart;
  int n = 0, p = 0;
  for (; p < rows + width - 1; p++) {
    plane_start[p] = n;
    for (int i = 0; i < rows; i++) {
      int j = p - i;
      if (j < 0 || j >= widths[i / height]) continue;
      order[n++] = i * width + j;
    }
  }
  plane_start[p] = n;
  return PCONV_OK;
}

namespace {
struct HaloEntry {
  int32_t plane, dst, src0, src1;
  float wgt;
};

// Causal version of the facing-column rule (entropy_context_cuda.cu:141-157):
// column i of a tile of width wfrom looks at columns c, c+1 of the neighbouring
// tile (width wto) with weights w, 1-w, clamped so that only already-coded
// columns contribute.  Returns false when nothing is visible yet.
inline bool causal_columns(int i, int wfrom, int wto, int *c_out, float *w_out) {
  float p = facing_column(i, wfrom, wto, false);
  int c = p < 0 ? -1 : static_cast<int>(p);
  float w;
  if (c > i) return false;
  if (c + 1 > i) {
    w = 1.f;
  } else {
    w = c + 1 - p;
    if (c == -1) w = 0.f;
  }
  *c_out = c;
  *w_out = w;
  return true;
}
}  // namespace

extern "C" int pconv_host_causal_table(const int32_t *widths, int npart, int height, int width, int pad,
                                       int32_t *col, float *wgt) {
  PCONV_REQUIRE(widths && col && wgt && pad >= 1, "causal_table: bad argument");
  const int rows = height * npart;
  for (int t = 0; t < npart; t++)
    for (int side = 0; side < 2; side++)
      for (int r = 0; r < pad; r++) {
        const int row = (side == 0) ? t * height - pad + r : (t + 1) * height + r;
        const size_t base = ((size_t)(t * 2 + side) * pad + r) * width;
        for (int i = 0; i < width; i++) {
          col[base + i] = -2;
          wgt[base + i] = 0.f;
          if (row < 0 || row >= rows || i >= widths[t]) continue;
          int c;
          float w;
          if (!causal_columns(i, widths[t], widths[row / height], &c, &w)) continue;
          col[base + i] = c;
          wgt[base + i] = w;
        }
      }
  return PCONV_OK;
}

// Table of the training-time causal pad (PseudoEntropyContextOp, pseudo_entropy_context_cuda.cu:51-170):
// same layout as pconv_host_causal_table.  version 1 is that very table; version 0 keeps both taps unless the
// next source column lies at or right of the destination column on the full-width grid.
extern "C" int pconv_host_entropy_pad_table(const int32_t *widths, int npart, int height, int width, int pad,
                                            int version, int32_t *col, float *wgt) {
  PCONV_REQUIRE(version == 0 || version == 1, "entropy_pad_table: undefined context version");
  if (version == 1) return pconv_host_causal_table(widths, npart, height, width, pad, col, wgt);
  PCONV_REQUIRE(widths && col && wgt && pad >= 1, "entropy_pad_table: bad argument");
  const int rows = height * npart;
  for (int t = 0; t < npart; t++)
    for (int side = 0; side < 2; side++)
      for (int r = 0; r < pad; r++) {
        const int row = (side == 0) ? t * height - pad + r : (t + 1) * height + r;
        const size_t base = ((size_t)(t * 2 + side) * pad + r) * width;
        for (int i = 0; i < width; i++) {
          col[base + i] = -2;
          wgt[base + i] = 0.f;
          if (row < 0 || row >= rows || i >= widths[t]) continue;
          const int ws = widths[row / height];
          const float p = facing_column(i, widths[t], ws, false);
          const int c = p < 0 ? -1 : static_cast<int>(p);
          float w = c + 1 - p;
          const float next_full = (c + 1 + 0.5) / ws * width - 0.5;
          const float here_full = (i + 0.5) / widths[t] * width - 0.5;
          if (next_full >= static_cast<int>(here_full) + 0.999)
            w = 1.f;
          else if (c == -1)
            w = 0.f;
          col[base + i] = c;
          wgt[base + i] = w;
        }
      }
  return PCONV_OK;
}

extern "C" int pconv_host_causal_halo(const int32_t *widths, int npart, int channel,
                                      int height, int width, int pad, int32_t *dst,
                                      int32_t *src0, int32_t *src1, float *wgt,
                                      int32_t *entry_plane, int32_t *plane_start) {
  PCONV_REQUIRE(widths && plane_start, "causal_halo: null pointer");
  PCONV_REQUIRE(channel >= 1 && pad >= 1, "causal_halo: bad channel/pad");
  const int rows = height * npart;
  const int ph = height + 2 * pad, pw = width + 2 * pad;
  const long long tile_stride = (long long)channel * ph * pw;
  PCONV_REQUIRE(tile_stride * npart < (1LL << 31), "causal_halo: image exceeds int32 offsets");
  const int nplane = rows + width + pad - 1;
  std::vector<std::vector<HaloEntry>> lists(nplane);
  // vertical halo: lerp from the facing columns of the neighbouring tile,
  // clamped so that only already-coded columns contribute.
  for (int t = 0; t < npart; t++)
    for (int side = 0; side < 2; side++)
      for (int r = 0; r < pad; r++) {
        int row = (side == 0) ? t * height - pad + r : (t + 1) * height + r;
        if (row < 0 || row >= rows) continue;  // no pole mirroring in the causal model
        const int st = row / height;
        const int drow = (side == 0) ? r : pad + height + r;
        const int32_t dbase = (int32_t)(t * tile_stride + (long long)drow * pw);
        const int32_t sbase = (int32_t)(st * tile_stride + (long long)(pad + row % height) * pw);
        for (int i = 0; i < widths[t]; i++) {
          int c;
          float w;
          if (!causal_columns(i, widths[t], widths[st], &c, &w)) continue;  // halo stays zero
          HaloEntry e;
          e.plane = row + i;
          e.dst = dbase + i + pad;
          e.src0 = (c < 0) ? -1 : sbase + c + pad;
          e.src1 = sbase + (c + 1) % widths[st] + pad;
          e.wgt = w;
          lists[e.plane].push_back(e);
        }
      }
  // right-hand wrap: the first `pad` valid columns re-appear after the last one
  for (int t = 0; t < npart; t++)
    for (int r = 0; r < ph; r++) {
      int row = t * height + r - pad;
      if (row < 0 || row >= rows) continue;
      for (int i = 0; i < pad; i++) {
        HaloEntry e;
        e.plane = row + i + widths[t];
        int32_t base = (int32_t)(t * tile_stride + (long long)r * pw + i + pad);
        e.dst = base + widths[t];
        e.src0 = base;
        e.src1 = -2;
        e.wgt = 1.f;
        lists[e.plane].push_back(e);
      }
    }
  int n = 0;
  for (int p = 0; p < nplane; p++) {
    plane_start[p] = n;
    if (dst) {
      for (const HaloEntry &e : lists[p]) {
        dst[n] = e.dst;
        src0[n] = e.src0;
        src1[n] = e.src1;
        wgt[n] = e.wgt;
        entry_plane[n] = e.plane;
        n++;
      }
    } else {
      n += (int)lists[p].size();
    }
  }
  plane_start[nplane] = n;
  return n;
}

// Rodrigues rotation matrix of axis*angle vector v (projects_cuda.cu:20-49)
static void rodrigues(float x, float y, float z, float *m) {
  float norm = sqrt(x * x + y * y + z * z);
  for (int i = 0; i < 9; i++) m[i] = 0;
  if (norm == 0) {
    m[0] = m[4] = m[8] = 1.f;
    return;
  }
  float tx = x / norm, ty = y / norm, tz = z / norm;
  float c = cos(norm), s = sin(norm);
  m[0] = c + (1 - c) * tx * tx;
  m[1] = (1 - c) * tx * ty - s * tz;
  m[2] = (1 - c) * tx * tz + s * ty;
  m[3] = (1 - c) * ty * tx + s * tz;
  m[4] = c + (1 - c) * ty * ty;
  m[5] = (1 - c) * ty * tz - s * tx;
  m[6] = (1 - c) * tz * tx - s * ty;
  m[7] = (1 - c) * tz * ty + s * tx;
  m[8] = c + (1 - c) * tz * tz;
}

extern "C" int pconv_host_project_table(const float *theta, const float *phi, int nview,
                                        float fov_in, int h_out, int w_out, int height,
                                        int width, float *tf) {
  PCONV_REQUIRE(theta && phi && tf && nview > 0, "project_table: bad argument");
  const float pi = acos(-1.0);
  // the reference scales its angles by pi in the constructor (projects.hpp)
  const float fov = fov_in * pi;
  float hfov = fov * h_out / w_out / 2;
  float wfov = fov / 2;
  float cx = (w_out - 1) / 2.0;
  float cy = (h_out - 1) / 2.0;
  float pi_2 = pi / 2;
  float w_stride = 2 * sin(wfov) / sin(pi_2 - wfov) / (w_out - 1);
  float h_stride = 2 * sin(hfov) / sin(pi_2 - hfov) / (h_out - 1);
  float hx = (width - 1) / 2.0;
  float hy = (height - 1) / 2.0;
  const int inner = h_out * w_out;
  for (int v = 0; v < nview; v++) {
    float r1[9], r2[9], r[9];
    rodrigues(0, 0, theta[v] * pi, r1);
    float nphi = -(phi[v] * pi);
    rodrigues(r1[1] * nphi, r1[4] * nphi, r1[7] * nphi, r2);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        float s = 0;
        for (int k = 0; k < 3; k++) s += r2[a * 3 + k] * r1[k * 3 + b];
        r[a * 3 + b] = s;
      }
    for (int i = 0; i < inner; i++) {
      int w = i % w_out, h = i / w_out;
      float x = 1.;
      float y = (w - cx) * w_stride;
      float z = (h - cy) * h_stride;
      float len = sqrt(x * x + y * y + z * z);
      float xa = x / len, xb = y / len, xc = -z / len;
      float px = xa * r[0] + xb * r[1] + xc * r[2];
      float py = xa * r[3] + xb * r[4] + xc * r[5];
      float pz = xa * r[6] + xb * r[7] + xc * r[8];
      float lat = asin(pz);
      float th = atan(py / px);
      if (px <= 0) th = (py > 0) ? th + pi : th - pi;
      size_t e = ((size_t)v * inner + i) * 2;
      tf[e] = th / pi * hx + hx;
      tf[e + 1] = -2 * lat / pi * hy + hy;
    }
  }
  return PCONV_OK;
}
