// Native entropy engine: the EntEncoder / EntDecoder wavefront loops of the
// reference (pseudo_codec.py:97-114, 145-160) as a C++ host loop over the SAME
// step kernels the PCONV op classes launch (entropy.hip), so a stream written by
// either path decodes with the other.
//
// What changes against the per-op Python path is only the orchestration:
//   * one host loop in C++, kernels launched back to back on one stream;
//   * frames of a batch run in lock-step (`nimg`), one arithmetic coder each;
//   * only the live rows of a step travel over PCIe, through pinned buffers
//     (the reference moves a full (16h*w, 9) table and a full label plane per
//     step, pseudo_codec.py:112,157-158);
//   * the encoder never waits for the GPU inside the loop: tables and labels of
//     all steps are written to one device buffer in stream order and copied
//     once; the CPU coder then runs over it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/pconv_coder.h"
#include "common.h"

namespace {

constexpr int kLayers = 12;
constexpr int kPad = 2;
constexpr int kKernel = 5;

#define HIP_TRY(expr)                                                     \
  do {                                                                    \
    hipError_t e__ = (expr);                                              \
    if (e__ != hipSuccess) {                                              \
      pconv_set_error("engine: %s: %s", #expr, hipGetErrorString(e__));   \
      return PCONV_ELAUNCH;                                               \
    }                                                                     \
  } while (0)

#define PC_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ < 0) return rc__;  \
  } while (0)

struct HaloList {
  int32_t *dst = nullptr, *src0 = nullptr, *src1 = nullptr, *plane = nullptr;
  float *wgt = nullptr;
  std::vector<int32_t> start;  // per plane, host
};

struct Window {
  int lo, len;
};

}  // namespace

struct pconv_entropy_engine {
  int npart, ngroup, h, w, nimg, nstep_levels;
  float bias, total, beta;
  int rows, nsteps, cpn;  // cpn = gaussians = outputs per group of the last layer
  std::vector<int32_t> widths, sched_start;
  int32_t *widths_d = nullptr, *order_d = nullptr, *sched_start_d = nullptr;
  int32_t *vh_col = nullptr;  // dense causal halo table (pconv_host_causal_table)
  float *vh_wgt = nullptr;
  int longest_plane = 0;
  bool stored_halo = false;  // debugging aid: run EntropyCtxPadRun2 launches instead of virtual halos
  HaloList halo_in, halo_hid;
  const float *lw[kLayers] = {nullptr}, *lb[kLayers] = {nullptr}, *la[kLayers] = {nullptr};
  float *ctx = nullptr;             // (3*nimg*npart, ngroup, h+4, w+4)
  float *act[kLayers] = {nullptr};  // layer outputs, persistent across steps
  float *packed = nullptr;          // symbols of the previous step, [img][len]
  int32_t *tables_d = nullptr;      // step tables (decode) / all tables (encode)
  int32_t *labels_d = nullptr;
  int32_t *tables_h = nullptr, *labels_h = nullptr;  // pinned
  float *packed_h = nullptr;                         // pinned
  size_t sym_per_img = 0, max_len = 0;
  std::vector<std::vector<uint8_t>> streams;
  std::vector<pconv_coder *> coders;

  size_t ctx_elems() const { return (size_t)3 * nimg * npart * ngroup * (h + 2 * kPad) * (w + 2 * kPad); }
  size_t act_elems(int l) const {
    const int p = (l == kLayers - 1) ? 0 : kPad;
    return (size_t)3 * nimg * npart * 3 * ngroup * (h + 2 * p) * (w + 2 * p);
  }

  Window sched_window(int psum) const {
    int st = psum - ngroup + 1 < 0 ? 0 : psum - ngroup + 1;
    int end = psum < rows + w - 2 ? psum + 1 : rows + w - 1;
    if (st > end) return {0, 0};
    return {sched_start[st], sched_start[end] - sched_start[st]};
  }
  Window halo_window(const HaloList &hl, int psum) const {
    if (psum < 0 || psum >= rows + w + kPad + ngroup - 2) return {0, 0};
    int st = psum - ngroup + 1 < 0 ? 0 : psum - ngroup + 1;
    int end = psum < rows + w + kPad - 2 ? psum + 1 : rows + w + kPad - 1;
    return {hl.start[st], hl.start[end] - hl.start[st]};
  }

  int build_halo(HaloList &hl, int channel) {
    const int nplane = rows + w + kPad - 1;
    hl.start.assign(nplane + 1, 0);
    int n = pconv_host_causal_halo(widths.data(), npart, channel, h, w, kPad, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, hl.start.data());
    if (n < 0) return n;
    std::vector<int32_t> d(n + 1), s0(n + 1), s1(n + 1), pl(n + 1);
    std::vector<float> wg(n + 1);
    n = pconv_host_causal_halo(widths.data(), npart, channel, h, w, kPad, d.data(), s0.data(), s1.data(),
                               wg.data(), pl.data(), hl.start.data());
    if (n < 0) return n;
    const size_t bytes = (size_t)(n + 1) * 4;
    HIP_TRY(hipMalloc(&hl.dst, bytes));
    HIP_TRY(hipMalloc(&hl.src0, bytes));
    HIP_TRY(hipMalloc(&hl.src1, bytes));
    HIP_TRY(hipMalloc(&hl.plane, bytes));
    HIP_TRY(hipMalloc(&hl.wgt, bytes));
    HIP_TRY(hipMemcpy(hl.dst, d.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hl.src0, s0.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hl.src1, s1.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hl.plane, pl.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(hl.wgt, wg.data(), bytes, hipMemcpyHostToDevice));
    return PCONV_OK;
  }

  int init(const float *tile_weight) {
    rows = h * npart;
    nsteps = rows + w + ngroup - 2;
    cpn = 3;
    widths.assign(npart, 0);
    PC_TRY(pconv_host_tile_widths(tile_weight, npart, rows, w, widths.data()));
    std::vector<int32_t> order((size_t)rows * w);
    sched_start.assign(rows + w, 0);
    PC_TRY(pconv_host_wavefront(widths.data(), npart, h, w, order.data(), sched_start.data()));
    const size_t npos = sched_start[rows + w - 1];
    sym_per_img = npos * ngroup;
    max_len = 0;
    for (int s = 0; s < nsteps; s++) {
      Window wd = sched_window(s);
      if ((size_t)wd.len > max_len) max_len = wd.len;
    }
    HIP_TRY(hipMalloc(&widths_d, npart * 4));
    HIP_TRY(hipMemcpy(widths_d, widths.data(), npart * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&order_d, order.size() * 4));
    HIP_TRY(hipMemcpy(order_d, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&sched_start_d, sched_start.size() * 4));
    HIP_TRY(hipMemcpy(sched_start_d, sched_start.data(), sched_start.size() * 4, hipMemcpyHostToDevice));
    longest_plane = 0;
    for (int p = 0; p + 1 < rows + w; p++)
      if (sched_start[p + 1] - sched_start[p] > longest_plane) longest_plane = sched_start[p + 1] - sched_start[p];
    {
      const size_t n = (size_t)npart * 2 * kPad * w;
      std::vector<int32_t> col(n);
      std::vector<float> wg(n);
      PC_TRY(pconv_host_causal_table(widths.data(), npart, h, w, kPad, col.data(), wg.data()));
      HIP_TRY(hipMalloc(&vh_col, n * 4));
      HIP_TRY(hipMalloc(&vh_wgt, n * 4));
      HIP_TRY(hipMemcpy(vh_col, col.data(), n * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(vh_wgt, wg.data(), n * 4, hipMemcpyHostToDevice));
    }
    stored_halo = getenv("PCONV_ENGINE_STORED_HALO") != nullptr;
    if (stored_halo) {
      PC_TRY(build_halo(halo_in, ngroup));
      PC_TRY(build_halo(halo_hid, 3 * ngroup));
    }
    HIP_TRY(hipMalloc(&ctx, ctx_elems() * 4));
    for (int l = 0; l < kLayers; l++) HIP_TRY(hipMalloc(&act[l], act_elems(l) * 4));
    HIP_TRY(hipMalloc(&packed, (size_t)nimg * rows * w * 4));
    const size_t all_rows = sym_per_img * nimg;
    HIP_TRY(hipMalloc(&tables_d, all_rows * (nstep_levels + 1) * 4));
    HIP_TRY(hipMalloc(&labels_d, all_rows * 4));
    HIP_TRY(hipHostMalloc(&tables_h, all_rows * (nstep_levels + 1) * 4));
    HIP_TRY(hipHostMalloc(&labels_h, all_rows * 4));
    HIP_TRY(hipHostMalloc(&packed_h, (size_t)nimg * max_len * 4));
    streams.resize(nimg);
    for (int i = 0; i < nimg; i++) coders.push_back(pconv_coder_new(nullptr));
    return PCONV_OK;
  }

  void release() {
    auto freed = [](void *p) {
      if (p) (void)hipFree(p);
    };
    freed(widths_d); freed(order_d); freed(sched_start_d); freed(vh_col); freed(vh_wgt); freed(ctx); freed(packed); freed(tables_d); freed(labels_d);
    for (int l = 0; l < kLayers; l++) freed(act[l]);
    for (HaloList *hl : {&halo_in, &halo_hid}) {
      freed(hl->dst); freed(hl->src0); freed(hl->src1); freed(hl->plane); freed(hl->wgt);
    }
    if (tables_h) (void)hipHostFree(tables_h);
    if (labels_h) (void)hipHostFree(labels_h);
    if (packed_h) (void)hipHostFree(packed_h);
    for (pconv_coder *c : coders) pconv_coder_free(c);
  }

  int clear(hipStream_t st) {
    HIP_TRY(hipMemsetAsync(ctx, 0, ctx_elems() * 4, st));
    for (int l = 0; l < kLayers; l++) HIP_TRY(hipMemsetAsync(act[l], 0, act_elems(l) * 4, st));
    return PCONV_OK;
  }

  // one wavefront step of the three-headed network.  prev_symbols != NULL: scatter
  // the previous step's symbols first (decoder); the encoder fills ctx up front.
  int network_step(int s, Window prev, Window cur, const float *prev_symbols, hipStream_t st) {
    const int n3 = 3 * nimg;
    const int hid = 3 * ngroup;
    if (s > 0 && prev.len > 0 && prev_symbols)
      PC_TRY(pconv_dinput2(prev_symbols, ctx, order_d, prev.lo, prev.len, nimg, ngroup, npart, h, w, kPad, s - 1,
                           -bias, 3, st));
    const int first = s - ngroup + 1 < 0 ? 0 : s - ngroup + 1;
    const int end = s < rows + w - 2 ? s + 1 : rows + w - 1;
    for (int l = 0; l < kLayers; l++) {
      float *in = (l == 0) ? ctx : act[l - 1];
      const int channel = (l == 0) ? ngroup : hid;
      if (stored_halo) {
        const HaloList &hl = (l == 0) ? halo_in : halo_hid;
        const int psum_pad = (l == 0) ? s - 1 : s;  // the input layer lags one step
        Window hw = halo_window(hl, psum_pad);
        if (hw.len > 0)
          PC_TRY(pconv_ctx_pad_run2(in, hl.dst, hl.src0, hl.src1, hl.wgt, hl.plane, hw.lo, hw.len, n3,
                                    channel / ngroup, channel, npart, h, w, kPad, psum_pad, st));
      }
      if (cur.len > 0) {
        // second conv of a residual block: += block input, folded into the epilogue
        const float *res = (l >= 2 && l <= 10 && (l % 2) == 0) ? act[l - 2] : nullptr;
        PC_TRY(pconv_entropy_conv(in, lw[l], lb[l], la[l], act[l], order_d, sched_start_d, first, end - first,
                                  longest_plane, n3, nimg, channel, hid, ngroup, kKernel, l == 0 ? 5 : 6, npart,
                                  h, w, kPad, l == kLayers - 1 ? 0 : kPad, s, res,
                                  stored_halo ? nullptr : widths_d, stored_halo ? nullptr : vh_col,
                                  stored_halo ? nullptr : vh_wgt, st));
      }
    }
    return PCONV_OK;
  }

  int step_tables(int s, Window cur, const float *symbols, int32_t *table_out, int32_t *labels_out,
                  hipStream_t st) {
    if (cur.len <= 0) return PCONV_OK;
    return pconv_step_tables(act[kLayers - 1], symbols, table_out, labels_out, order_d, cur.lo, cur.len, nimg,
                             ngroup, npart, h, w, s, nstep_levels, bias, total, beta, st);
  }
};

namespace {

template <typename Fn>
void for_each_image(int nimg, Fn fn) {
  if (nimg == 1) {
    fn(0);
    return;
  }
  std::vector<std::thread> pool;
  for (int i = 1; i < nimg; i++) pool.emplace_back(fn, i);
  fn(0);
  for (std::thread &t : pool) t.join();
}

}  // namespace

extern "C" {

pconv_entropy_engine *pconv_ee_create(int npart, int ngroup, int h, int w, int nimg, const float *tile_weight,
                                      float bias, int nlevels, float total, float beta) {
  if (npart <= 0 || ngroup <= 0 || h <= 0 || w <= 0 || nimg <= 0 || !tile_weight || nlevels <= 0) {
    pconv_set_error("ee_create: bad argument");
    return nullptr;
  }
  pconv_entropy_engine *e = new pconv_entropy_engine();
  e->npart = npart; e->ngroup = ngroup; e->h = h; e->w = w; e->nimg = nimg;
  e->bias = bias; e->nstep_levels = nlevels; e->total = total; e->beta = beta;
  if (e->init(tile_weight) != PCONV_OK) {
    e->release();
    delete e;
    return nullptr;
  }
  return e;
}

void pconv_ee_destroy(pconv_entropy_engine *e) {
  if (!e) return;
  e->release();
  delete e;
}

// device pointers in the reference's parameter layout: weight (3, 3G, cin, 5, 5),
// bias (3, 3G), slope (3, 3G) or NULL (EntropyContextNew.py:245-249)
int pconv_ee_set_layer(pconv_entropy_engine *e, int layer, const float *weight, const float *bias,
                       const float *slope) {
  PCONV_REQUIRE(e && layer >= 0 && layer < kLayers && weight && bias, "ee_set_layer: bad argument");
  e->lw[layer] = weight;
  e->lb[layer] = bias;
  e->la[layer] = slope;
  return PCONV_OK;
}

long long pconv_ee_symbols_per_image(const pconv_entropy_engine *e) { return e ? (long long)e->sym_per_img : -1; }
int pconv_ee_steps(const pconv_entropy_engine *e) { return e ? e->nsteps : -1; }

// symbols: device float (nimg*npart, ngroup, h, w), dead columns already zeroed
// (PseudoFill).  Streams are kept inside the engine (pconv_ee_stream).
int pconv_ee_encode(pconv_entropy_engine *e, const float *symbols, void *stream) {
  PCONV_REQUIRE(e && symbols, "ee_encode: bad argument");
  for (int l = 0; l < kLayers; l++) PCONV_REQUIRE(e->lw[l], "ee_encode: layer %d has no weights", l);
  hipStream_t st = as_stream(stream);
  const int cols = e->nstep_levels + 1;
  PC_TRY(e->clear(st));
  PC_TRY(pconv_symbols_to_ctx(symbols, e->ctx, e->widths_d, e->nimg * e->npart, e->ngroup, e->h, e->w, kPad,
                              e->npart, -e->bias, 3, st));
  Window prev = {0, 0};
  size_t row = 0;  // rows are laid out [step][img][l]
  std::vector<size_t> step_row(e->nsteps + 1, 0);
  for (int s = 0; s < e->nsteps; s++) {
    Window cur = e->sched_window(s);
    step_row[s] = row;
    PC_TRY(e->network_step(s, prev, cur, nullptr, st));
    PC_TRY(e->step_tables(s, cur, symbols, e->tables_d + row * cols, e->labels_d + row, st));
    row += (size_t)cur.len * e->nimg;
    prev = cur;
  }
  step_row[e->nsteps] = row;
  HIP_TRY(hipMemcpyAsync(e->tables_h, e->tables_d, row * cols * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(e->labels_h, e->labels_d, row * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int status = 0;
  for_each_image(e->nimg, [&](int img) {
    pconv_coder *c = e->coders[img];
    int rc = pconv_coder_start_encoder(c);
    for (int s = 0; s < e->nsteps && rc >= 0; s++) {
      const size_t len = (step_row[s + 1] - step_row[s]) / e->nimg;
      if (!len) continue;
      const size_t r0 = step_row[s] + (size_t)img * len;
      rc = pconv_coder_encodes(c, e->tables_h + r0 * cols, e->nstep_levels, e->labels_h + r0, (int)len);
    }
    if (rc >= 0) rc = pconv_coder_end_encoder(c);
    if (rc < 0) {
      status = rc;
      pconv_set_error("ee_encode: coder of image %d: %s", img, pconv_coder_error(c));
      return;
    }
    size_t nb = 0;
    const uint8_t *p = pconv_coder_bytes(c, &nb);
    e->streams[img].assign(p, p + nb);
  });
  return status < 0 ? PCONV_EINVAL : PCONV_OK;
}

const uint8_t *pconv_ee_stream(const pconv_entropy_engine *e, int img, size_t *nbytes) {
  if (!e || img < 0 || img >= e->nimg) return nullptr;
  if (nbytes) *nbytes = e->streams[img].size();
  return e->streams[img].data();
}

// streams[img] / nbytes[img]: host buffers.  symbols_out: device float
// (nimg*npart, ngroup, h, w) = decoded indices, zeros in the dead columns.
int pconv_ee_decode(pconv_entropy_engine *e, const uint8_t *const *streams, const size_t *nbytes,
                    float *symbols_out, void *stream) {
  PCONV_REQUIRE(e && streams && nbytes && symbols_out, "ee_decode: bad argument");
  for (int l = 0; l < kLayers; l++) PCONV_REQUIRE(e->lw[l], "ee_decode: layer %d has no weights", l);
  hipStream_t st = as_stream(stream);
  const int cols = e->nstep_levels + 1;
  for (int i = 0; i < e->nimg; i++)
    if (pconv_coder_start_decoder_mem(e->coders[i], streams[i], nbytes[i]) < 0) {
      pconv_set_error("ee_decode: cannot start decoder %d", i);
      return PCONV_EINVAL;
    }
  PC_TRY(e->clear(st));
  Window prev = {0, 0};
  std::vector<int32_t> sym;
  int status = 0;
  for (int s = 0; s < e->nsteps; s++) {
    Window cur = e->sched_window(s);
    PC_TRY(e->network_step(s, prev, cur, e->packed, st));
    if (cur.len > 0) {
      const size_t rows = (size_t)cur.len * e->nimg;
      PC_TRY(e->step_tables(s, cur, nullptr, e->tables_d, nullptr, st));
      HIP_TRY(hipMemcpyAsync(e->tables_h, e->tables_d, rows * cols * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      sym.resize(rows);
      for_each_image(e->nimg, [&](int img) {
        int rc = pconv_coder_decodes_i32(e->coders[img], e->tables_h + (size_t)img * cur.len * cols,
                                         e->nstep_levels, sym.data() + (size_t)img * cur.len, cur.len);
        if (rc < 0) status = rc;
      });
      if (status < 0) {
        pconv_set_error("ee_decode: arithmetic decoder desynchronised at step %d", s);
        return PCONV_EINVAL;
      }
      for (size_t i = 0; i < rows; i++) e->packed_h[i] = (float)sym[i];
      HIP_TRY(hipMemcpyAsync(e->packed, e->packed_h, rows * 4, hipMemcpyHostToDevice, st));
    }
    prev = cur;
  }
  // the last planes were never scattered by a following step: do it now, then
  // read the symbols back out of the context tensor (pseudo_codec.py:159)
  if (prev.len > 0)
    PC_TRY(pconv_dinput2(e->packed, e->ctx, e->order_d, prev.lo, prev.len, e->nimg, e->ngroup, e->npart, e->h,
                         e->w, kPad, e->nsteps - 1, -e->bias, 1, st));
  PC_TRY(pconv_ctx_to_symbols(e->ctx, symbols_out, e->widths_d, e->nimg * e->npart, e->ngroup, e->h, e->w, kPad,
                              e->npart, e->bias, st));
  return PCONV_OK;
}

}  // extern "C"
