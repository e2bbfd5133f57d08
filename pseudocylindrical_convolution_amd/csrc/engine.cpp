// Native entropy engine: the EntEncoder / EntDecoder wavefront loops of the
// reference (pseudo_codec.py:97-114, 145-160) as a C++ host loop.
//
// Arithmetic is the per-op path's (PCONV.EntropyConv2Op & co, entropy.hip): same
// causal masks, same reduction order, same CDF construction, so a stream written
// by either path decodes with the other (tests/test_gpu_engine.py).  What differs
// is orchestration and layout:
//   * one host loop in C++, per step 1 scatter + 12 layer launches + 1 table
//     launch (the per-op path: 12 halo updates, 12 convs, 5 adds, 2 extracts,
//     4 GMM launches, driven from Python);
//   * engine-private channels-last buffers whose halos are written by the
//     producer of the value they derive from (entropy_engine.hip);
//   * `nimg` frames advance in lock-step, one arithmetic coder each;
//   * only the live rows of a step cross PCIe, through pinned buffers; the
//     encoder never waits inside the loop: tables and labels of all steps are
//     written to one device buffer in stream order, copied once and coded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/pconv_coder.h"
#include "common.h"
#include "ee_kernels.h"

namespace {

constexpr int kLayers = 12;
constexpr int kPad = 2;

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

struct Window {
  int lo, len, first, nplane;
};

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

struct pconv_entropy_engine {
  int npart, ngroup, h, w, nimg, nlevels;
  float bias, total, beta;
  int rows, nsteps, longest_plane = 0;
  std::vector<int32_t> widths, sched_start;
  EeGeom geom;
  int32_t *widths_d = nullptr, *order_d = nullptr, *sched_start_d = nullptr, *vh_col = nullptr;
  int32_t *rev_start_d = nullptr, *rev_entry_d = nullptr;
  float *vh_wgt = nullptr;
  int32_t *bulk_wg_d = nullptr, *pos_plane_d = nullptr, *step_row_d = nullptr;
  std::vector<int32_t> step_row;  // first table row of each step (x nimg), +1 end
  bool stepwise_encoder = false;  // debugging aid: encode step by step like the decoder
  float *lw[kLayers] = {nullptr};  // engine-owned packed weights
  const float *lb[kLayers] = {nullptr}, *la[kLayers] = {nullptr};
  bool bound[kLayers] = {false};
  float *ctx = nullptr;             // (nimg*npart, h+4, w+4, G)
  float *act[kLayers] = {nullptr};  // (3*nimg*npart, h+2p, w+2p, 3G), persistent across steps
  float *packed = nullptr;          // decoder: symbols of the previous step [img][len]
  int32_t *tables_d = nullptr, *labels_d = nullptr;
  int32_t *tables_h = nullptr, *labels_h = nullptr;  // pinned
  float *packed_h = nullptr;                         // pinned
  size_t sym_per_img = 0, max_len = 0;
  std::vector<std::vector<uint8_t>> streams;
  std::vector<pconv_coder *> coders;

  size_t ctx_elems() const { return (size_t)nimg * npart * ngroup * (h + 2 * kPad) * (w + 2 * kPad); }
  size_t act_elems(int l) const {
    const int p = (l == kLayers - 1) ? 0 : kPad;
    return (size_t)3 * nimg * npart * 3 * ngroup * (h + 2 * p) * (w + 2 * p);
  }
  int layer_cin(int l) const { return l == 0 ? ngroup : 3 * ngroup; }

  Window window(int psum) const {
    int st = psum - ngroup + 1 < 0 ? 0 : psum - ngroup + 1;
    int end = psum < rows + w - 2 ? psum + 1 : rows + w - 1;
    if (st >= end) return {0, 0, 0, 0};
    return {sched_start[st], sched_start[end] - sched_start[st], st, end - st};
  }

  int init(const float *tile_weight) {
    rows = h * npart;
    nsteps = rows + w + ngroup - 2;
    widths.assign(npart, 0);
    PC_TRY(pconv_host_tile_widths(tile_weight, npart, rows, w, widths.data()));
    std::vector<int32_t> order((size_t)rows * w);
    sched_start.assign(rows + w, 0);
    PC_TRY(pconv_host_wavefront(widths.data(), npart, h, w, order.data(), sched_start.data()));
    sym_per_img = (size_t)sched_start[rows + w - 1] * ngroup;
    for (int p = 0; p + 1 < rows + w; p++)
      if (sched_start[p + 1] - sched_start[p] > longest_plane) longest_plane = sched_start[p + 1] - sched_start[p];
    for (int s = 0; s < nsteps; s++)
      if ((size_t)window(s).len > max_len) max_len = window(s).len;
    HIP_TRY(hipMalloc(&widths_d, npart * 4));
    HIP_TRY(hipMemcpy(widths_d, widths.data(), npart * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&order_d, order.size() * 4));
    HIP_TRY(hipMemcpy(order_d, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&sched_start_d, sched_start.size() * 4));
    HIP_TRY(hipMemcpy(sched_start_d, sched_start.data(), sched_start.size() * 4, hipMemcpyHostToDevice));
    {
      const size_t n = (size_t)npart * 2 * kPad * w;
      std::vector<int32_t> col(n);
      std::vector<float> wg(n);
      PC_TRY(pconv_host_causal_table(widths.data(), npart, h, w, kPad, col.data(), wg.data()));
      HIP_TRY(hipMalloc(&vh_col, n * 4));
      HIP_TRY(hipMalloc(&vh_wgt, n * 4));
      HIP_TRY(hipMemcpy(vh_col, col.data(), n * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(vh_wgt, wg.data(), n * 4, hipMemcpyHostToDevice));
      // reverse map: interior (global row, column) -> halo entries interpolated from it
      // (entry en = ((tile*2 + side)*kPad + r)*w + column reads source columns c and
      // c+1, circular, of the neighbouring tile's row; entry columns past the tile's
      // valid width are never read and stay out)
      std::vector<std::vector<int32_t>> lists((size_t)rows * w);
      for (size_t en = 0; en < n; en++) {
        const int cp = (int)(en % w);
        size_t q = en / w;
        const int r = (int)(q % kPad);
        q /= kPad;
        const int side = (int)(q & 1), tg = (int)(q >> 1);
        const int c = col[en];
        if (c == -2 || cp >= widths[tg]) continue;
        const int srow = side ? (tg + 1) * h + r : tg * h - kPad + r;
        if (srow < 0 || srow >= rows) continue;
        const int wst = widths[srow / h];
        int c1 = c + 1;
        c1 = c1 >= wst ? c1 - wst : c1;
        if (c >= 0) lists[(size_t)srow * w + c].push_back((int32_t)en);
        if (c1 != c) lists[(size_t)srow * w + c1].push_back((int32_t)en);
      }
      std::vector<int32_t> rstart(lists.size() + 1, 0), rentry;
      for (size_t k = 0; k < lists.size(); k++) {
        rentry.insert(rentry.end(), lists[k].begin(), lists[k].end());
        rstart[k + 1] = (int32_t)rentry.size();
      }
      if (rentry.empty()) rentry.push_back(0);
      HIP_TRY(hipMalloc(&rev_start_d, rstart.size() * 4));
      HIP_TRY(hipMalloc(&rev_entry_d, rentry.size() * 4));
      HIP_TRY(hipMemcpy(rev_start_d, rstart.data(), rstart.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(rev_entry_d, rentry.data(), rentry.size() * 4, hipMemcpyHostToDevice));
    }
    geom = {npart, ngroup, h, w, nimg, widths_d, order_d, sched_start_d, vh_col, vh_wgt, rev_start_d, rev_entry_d,
            nullptr, 0, nullptr, nullptr, 0};
    {  // bulk (encoder) maps
      const int npos = sched_start[rows + w - 1];
      std::vector<int32_t> wg, pp(npos);
      for (int p = 0; p + 1 < rows + w; p++) {
        const int cnt = sched_start[p + 1] - sched_start[p];
        for (int i = 0; i < cnt; i++) pp[sched_start[p] + i] = p;
        for (int f = 0; f < cnt; f += kEeBulkPos) {
          wg.push_back(p);
          wg.push_back(f);
        }
      }
      step_row.assign(nsteps + 1, 0);
      for (int s = 0; s < nsteps; s++) step_row[s + 1] = step_row[s] + window(s).len * nimg;
      HIP_TRY(hipMalloc(&bulk_wg_d, wg.size() * 4));
      HIP_TRY(hipMemcpy(bulk_wg_d, wg.data(), wg.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&pos_plane_d, pp.size() * 4));
      HIP_TRY(hipMemcpy(pos_plane_d, pp.data(), pp.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&step_row_d, step_row.size() * 4));
      HIP_TRY(hipMemcpy(step_row_d, step_row.data(), step_row.size() * 4, hipMemcpyHostToDevice));
      geom.bulk_wg = bulk_wg_d;
      geom.nbulk_wg = (int)(wg.size() / 2);
      geom.pos_plane = pos_plane_d;
      geom.step_row = step_row_d;
      geom.npos = npos;
      stepwise_encoder = getenv("PCONV_ENGINE_STEPWISE_ENCODER") != nullptr;
    }
    HIP_TRY(hipMalloc(&ctx, ctx_elems() * 4));
    for (int l = 0; l < kLayers; l++) {
      HIP_TRY(hipMalloc(&act[l], act_elems(l) * 4));
      HIP_TRY(hipMalloc(&lw[l], ee_packed_floats(3, 3 * ngroup, layer_cin(l)) * 4));
    }
    HIP_TRY(hipMalloc(&packed, (size_t)nimg * max_len * 4));
    const size_t all_rows = sym_per_img * nimg;
    HIP_TRY(hipMalloc(&tables_d, all_rows * (nlevels + 1) * 4));
    HIP_TRY(hipMalloc(&labels_d, all_rows * 4));
    HIP_TRY(hipHostMalloc(&tables_h, all_rows * (nlevels + 1) * 4));
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
    freed(widths_d); freed(order_d); freed(sched_start_d); freed(vh_col); freed(vh_wgt);
    freed(ctx); freed(packed); freed(tables_d); freed(labels_d);
    freed(bulk_wg_d); freed(pos_plane_d); freed(step_row_d); freed(rev_start_d); freed(rev_entry_d);
    for (int l = 0; l < kLayers; l++) {
      freed(act[l]);
      freed(lw[l]);
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

  // the 12 layers of one wavefront step (EntropyConvDBT / EntropyResidualBlockDBT
  // of pseudo_codec.py:27-51, 79-87)
  int network_step(int s, const Window &cur, hipStream_t st) {
    if (cur.len <= 0) return PCONV_OK;
    const int hid = 3 * ngroup;
    for (int l = 0; l < kLayers; l++) {
      const float *in = (l == 0) ? ctx : act[l - 1];
      // second conv of a residual block: += block input, folded into the epilogue
      const float *res = (l >= 2 && l <= 10 && (l % 2) == 0) ? act[l - 2] : nullptr;
      PC_TRY(ee_conv(&geom, in, l == 0, lw[l], lb[l], la[l], res, act[l], layer_cin(l), hid, l == 0 ? 5 : 6,
                     l == kLayers - 1 ? 0 : kPad, cur.first, cur.nplane, longest_plane, s, st));
    }
    return PCONV_OK;
  }

  // every layer once over all (plane, group) pairs: the encoder knows all symbols
  int network_bulk(hipStream_t st) {
    const int hid = 3 * ngroup;
    for (int l = 0; l < kLayers; l++) {
      const float *in = (l == 0) ? ctx : act[l - 1];
      const float *res = (l >= 2 && l <= 10 && (l % 2) == 0) ? act[l - 2] : nullptr;
      PC_TRY(ee_conv_bulk(&geom, in, l == 0, lw[l], lb[l], la[l], res, act[l], layer_cin(l), hid, l == 0 ? 5 : 6,
                          l == kLayers - 1 ? 0 : kPad, st));
      if (l != kLayers - 1) PC_TRY(ee_halo_bulk(&geom, act[l], hid, 3 * nimg, st));
    }
    return PCONV_OK;
  }
};

extern "C" {

pconv_entropy_engine *pconv_ee_create(int npart, int ngroup, int h, int w, int nimg, const float *tile_weight,
                                      float bias, int nlevels, float total, float beta) {
  if (npart <= 0 || ngroup <= 0 || h <= 0 || w <= 0 || nimg <= 0 || !tile_weight || nlevels <= 0) {
    pconv_set_error("ee_create: bad argument");
    return nullptr;
  }
  pconv_entropy_engine *e = new pconv_entropy_engine();
  e->npart = npart; e->ngroup = ngroup; e->h = h; e->w = w; e->nimg = nimg;
  e->bias = bias; e->nlevels = nlevels; e->total = total; e->beta = beta;
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
// bias (3, 3G), slope (3, 3G) or NULL (EntropyContextNew.py:245-249).  The weight
// is re-packed into the engine's reduction order; bias and slope are borrowed.
int pconv_ee_set_layer(pconv_entropy_engine *e, int layer, const float *weight, const float *bias,
                       const float *slope, void *stream) {
  PCONV_REQUIRE(e && layer >= 0 && layer < kLayers && weight && bias, "ee_set_layer: bad argument");
  PC_TRY(ee_pack_weight(weight, e->lw[layer], 3, 3 * e->ngroup, e->layer_cin(layer), stream));
  e->lb[layer] = bias;
  e->la[layer] = slope;
  e->bound[layer] = true;
  return PCONV_OK;
}

long long pconv_ee_symbols_per_image(const pconv_entropy_engine *e) { return e ? (long long)e->sym_per_img : -1; }
int pconv_ee_steps(const pconv_entropy_engine *e) { return e ? e->nsteps : -1; }

// symbols: device float (nimg*npart, ngroup, h, w) quantiser indices (dead
// columns are ignored).  Streams are kept inside the engine (pconv_ee_stream).
int pconv_ee_encode(pconv_entropy_engine *e, const float *symbols, void *stream) {
  PCONV_REQUIRE(e && symbols, "ee_encode: bad argument");
  for (int l = 0; l < kLayers; l++) PCONV_REQUIRE(e->bound[l], "ee_encode: layer %d has no weights", l);
  hipStream_t st = as_stream(stream);
  const int cols = e->nlevels + 1;
  // all symbols are known: fill the context once; the causal masks keep every
  // step from seeing more than DInput2 would have given it
  PC_TRY(e->clear(st));
  PC_TRY(ee_fill_ctx(&e->geom, symbols, e->ctx, -e->bias, st));
  PC_TRY(ee_halo_bulk(&e->geom, e->ctx, e->ngroup, e->nimg, st));
  const std::vector<int32_t> &step_row = e->step_row;  // rows are laid out [step][img][l]
  const size_t row = step_row[e->nsteps];
  if (e->stepwise_encoder) {
    for (int s = 0; s < e->nsteps; s++) {
      const Window cur = e->window(s);
      PC_TRY(e->network_step(s, cur, st));
      PC_TRY(ee_tables(&e->geom, e->act[kLayers - 1], symbols, e->tables_d + (size_t)step_row[s] * cols,
                       e->labels_d + step_row[s], cur.lo, cur.len, s, e->nlevels, e->bias, e->total, e->beta, st));
    }
  } else {
    PC_TRY(e->network_bulk(st));
    PC_TRY(ee_tables_bulk(&e->geom, e->act[kLayers - 1], symbols, e->tables_d, e->labels_d, e->nlevels, e->bias,
                          e->total, e->beta, st));
  }
  HIP_TRY(hipMemcpyAsync(e->tables_h, e->tables_d, row * cols * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(e->labels_h, e->labels_d, row * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int status = 0;
  for_each_image(e->nimg, [&](int img) {
    pconv_coder *c = e->coders[img];
    int rc = pconv_coder_start_encoder(c);
    for (int s = 0; s < e->nsteps && rc >= 0; s++) {
      const size_t len = (size_t)(step_row[s + 1] - step_row[s]) / e->nimg;
      if (!len) continue;
      const size_t r0 = (size_t)step_row[s] + (size_t)img * len;
      rc = pconv_coder_encodes(c, e->tables_h + r0 * cols, e->nlevels, e->labels_h + r0, (int)len);
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
  for (int l = 0; l < kLayers; l++) PCONV_REQUIRE(e->bound[l], "ee_decode: layer %d has no weights", l);
  hipStream_t st = as_stream(stream);
  const int cols = e->nlevels + 1;
  for (int i = 0; i < e->nimg; i++)
    if (pconv_coder_start_decoder_mem(e->coders[i], streams[i], nbytes[i]) < 0) {
      pconv_set_error("ee_decode: cannot start decoder %d", i);
      return PCONV_EINVAL;
    }
  PC_TRY(e->clear(st));
  Window prev = {0, 0, 0, 0};
  std::vector<int32_t> sym;
  int status = 0;
  for (int s = 0; s < e->nsteps; s++) {
    const Window cur = e->window(s);
    if (s > 0) PC_TRY(ee_scatter(&e->geom, e->packed, e->ctx, prev.lo, prev.len, s - 1, -e->bias, st));
    if (cur.len > 0) {
      const size_t rows = (size_t)cur.len * e->nimg;
      PC_TRY(e->network_step(s, cur, st));
      PC_TRY(ee_tables(&e->geom, e->act[kLayers - 1], nullptr, e->tables_d, nullptr, cur.lo, cur.len, s, e->nlevels,
                       e->bias, e->total, e->beta, st));
      HIP_TRY(hipMemcpyAsync(e->tables_h, e->tables_d, rows * cols * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      sym.resize(rows);
      for_each_image(e->nimg, [&](int img) {
        int rc = pconv_coder_decodes_i32(e->coders[img], e->tables_h + (size_t)img * cur.len * cols, e->nlevels,
                                         sym.data() + (size_t)img * cur.len, cur.len);
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
  // the symbols of the last step have not been scattered by a following step yet
  PC_TRY(ee_scatter(&e->geom, e->packed, e->ctx, prev.lo, prev.len, e->nsteps - 1, -e->bias, st));
  PC_TRY(ee_read_symbols(&e->geom, e->ctx, symbols_out, e->bias, st));
  return PCONV_OK;
}

}  // extern "C"
