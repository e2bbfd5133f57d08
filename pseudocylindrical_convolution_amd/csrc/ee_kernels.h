// Private interface between the native entropy engine (engine.cpp) and its
// channels-last kernels (entropy_engine.hip).  Not part of the public C ABI.
//
// Layout inside the engine ("tile-HWC"): a tensor with C channels is stored as
//   [image*npart + tile][row (+pad)][col (+pad)][C]
// so that the 5 x 5 x C window of a position is 5 runs of 5*C contiguous floats.
// The per-op API (include/pconv_hip.h) keeps the reference's NCHW.
//
// Halos are materialised by the PRODUCER of a value: whoever writes an interior
// element also rewrites the halo entries that are interpolated from it (reverse
// table rev_start / rev_entry) and, for the first two columns, their circular-wrap
// copies.  A halo entry therefore always equals a*t + b*(1-t) of the current
// values of its two source columns -- what a consumer evaluating
// pconv_host_causal_table on the fly would read -- and every position's window is
// a plain 5 x 5 x C block of the padded tile (no edge path in the consumers).
#pragma once
#include <stdint.h>

// One record per schedule entry (wavefront position), built once on the host: what a
// step kernel needs to find a position's window, output and halos without integer
// divisions (the scalar unit, one per CU, was the busiest resource of the step kernels).
struct EePos {
  int32_t pix;   // (tile*(h+4) + row)*(w+4) + col: origin of the 5 x 5 window in a buffer padded by 2;
                 // the position itself sits at pix + 2*(w+4) + 2 in such a buffer
  int32_t hw;    // (tile*h + row)*w + col: index in an unpadded buffer (= the schedule value)
  int32_t wrap;  // col < 2: valid width of the tile (the value is also stored `wrap` columns to the
                 // right, the circular wrap of the first columns); else 0
  int32_t rev;   // first reverse-halo record << 4 | count (rows that feed halos of neighbouring tiles)
};

// One record per (producing position, halo entry interpolated from it): the entry equals
// a*t + b*(1-t) of its two source columns, one of which is the producer.
struct EeHalo {
  int32_t dst;    // padded pixel index of the halo entry
  int32_t other;  // padded pixel index of the source that is NOT the producer (-1: none, reads as 0)
  float t;
  int32_t info;   // bits 0-15: wrap delta in columns (entry column < 2: also stored there), bit 30: the
                  // producer is the second source (b), bit 29: both sources are the producer
};

struct EeGeom {
  int npart, ngroup, h, w;  // tiles, channel groups, rows per tile, columns
  int nimg;                 // frames in lock-step (the network runs 3*nimg replicas)
  const int32_t *widths;    // device, valid width per tile
  const int32_t *order, *plane_start;  // device, wavefront schedule
  const int32_t *vh_col;    // device, dense causal halo table (bulk halo pass)
  const float *vh_wgt;
  const EePos *pos;         // device, one record per schedule entry
  const EeHalo *halo;       // device, reverse halo records (EePos::rev indexes them)
  const uint32_t *tap_in, *tap_hid;  // device, byte offset of reduction entry kk inside a window of the
                                     // context (ngroup channels) / of a hidden layer (3*ngroup), 0 past the end
  // bulk (encoder) mode: every (plane, group) pair at once
  const int32_t *pos_plane;  // device, plane of every schedule entry
  const int32_t *step_row;   // device, first table row of every step (rows are [step][img][l])
  int npos;
};

// reduction entries of a slab, padded to whole waves
static inline int ee_slab_slots(int cin) { return (cin * 25 + 63) / 64 * 64; }
// weights (3, cout, cin, 5, 5) -> per (set, output group) a slab [slot][4], slot = tap*cin + ci,
// holding the group's 3 rows interleaved (4th float: padding) with the CAUSAL MASK of the
// group applied (masked taps and the slots past the reduction length are zeros): tap (kh, kw)
// of input channel ci is usable by output group tc iff ci < (tc + slack + 4 - kh - kw)*(cin/ngroup),
// slack = 0 for the input layer (constrain 5), 1 for the hidden ones (constrain 6)
static inline size_t ee_packed_floats(int nset, int cout, int cin) {
  return (size_t)nset * (cout / 3) * ee_slab_slots(cin) * 4;
}
int ee_pack_weight(const float *w, float *packed, int nset, int cout, int cin, int ngroup, int constrain,
                   void *stream);

// one layer of one step.  x: cin channels, padded by 2; y: cout channels, padded
// by pad_out.  shared_input: x holds nimg images that every replica reads
// (layer 0), otherwise 3*nimg.  Output group of plane p is psum - p.
int ee_conv(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
            const float *slope, const float *residual, float *y, int cin, int cout, int constrain,
            int pad_out, int first_plane, int nplane, int longest_plane, int psum, void *stream);
// the same layer for ALL (plane, group) pairs at once (encoder: every symbol is
// known, the causal masks make each output equal to the step-by-step one)
int ee_conv_bulk(const EeGeom *g, const float *x, int shared_input, const float *packed_w, const float *bias,
                 const float *slope, const float *residual, float *y, int cin, int cout, int constrain,
                 int pad_out, int first_idx, int n_idx, int s_lo, int s_hi, void *stream);
// (first_idx, n_idx, s_lo, s_hi: only the (position, group) pairs of the wavefront steps [s_lo, s_hi), found among
// the schedule entries first_idx .. first_idx + n_idx - 1; the whole schedule: 0, npos, 0, INT_MAX)
// The same layer on the fp32 matrix cores (entropy_mfma.hip), identical bits: 42 -> 42 channels (14 groups),
// ee_mfma_block_shape: 1 and the block shape (nt rp_n rows x 16 ct_n columns per workgroup of
// `waves` waves, nt = 1 or 2 rows per wave) when the kernel takes the layer, else 0.  blocks: device int4 records (tile, first row, first
// column, 0) covering every live position; wfrag: ee_pack_weight_mfma's fragments (ee_mfma_packed_floats floats).
int ee_mfma_block_shape(int h, int cin, int *rp_n, int *ct_n, int *waves, int *nt);
int ee_mfma_packed_floats(int nset, int cin);
int ee_pack_weight_mfma(const float *w, float *packed, int nset, int cout, int cin, int ngroup, int constrain,
                        void *stream);
int ee_conv_bulk_mfma(const EeGeom *g, const void *blocks, int nblocks, int rp_n, int ct_n, int waves, int nt, const float *x,
                      int shared_input, const float *wfrag, const float *bias, const float *slope,
                      const float *residual, float *y, int cin, int cout, int pad_out, int s_lo, int s_hi,
                      void *stream);
// the same 42 -> 42 layer, four lane classes per instruction (v_mfma_f32_16x16x1_4b_f32), nt = 1 block shapes only
int ee_mfma4_packed_floats(int nset, int cin);
int ee_pack_weight_mfma4(const float *w, float *packed, int nset, int cout, int cin, int ngroup, int constrain,
                         void *stream);
int ee_conv_bulk_mfma4(const EeGeom *g, const void *blocks, int nblocks, int rp_n, int ct_n, int waves, const float *x,
                       const float *wfrag4, const float *bias, const float *slope, const float *residual, float *y,
                       int cin, int cout, int pad_out, int s_lo, int s_hi, void *stream);
// CDF rows and labels of the symbols of those steps, written in stream order
// packed != 0 (8 symbols, total 65536 only): `table` receives the coder's 16-byte rows (uint16 c1 .. c7 + label, see
// include/pconv_coder.h pconv_coder_encodes_rows16), `labels` is not written
int ee_tables_bulk(const EeGeom *g, const float *y_last, const float *symbols, int32_t *table, int32_t *labels,
                   int nstep, float bias, float total, float beta, int first_idx, int n_idx, int s_lo, int s_hi,
                   int packed, void *stream);

// decoder (r6): the LAST layer of a step (no activation, no residual, unpadded output nobody else reads) and the packed
// CDF rows of its positions in one launch; x = the last layer's input (3*ngroup channels, padded by 2), rows as
// ee_tables(packed) writes them.  The codec's shape only: 14 groups, 8 symbols, total 65536.
int ee_conv_tables(const EeGeom *g, const float *x, const float *packed_w, const float *bias, int32_t *table, int cin,
                   int first_plane, int nplane, int longest_plane, int psum, int lo, int len, float gbias, float total,
                   float beta, int32_t *counter, int32_t *flags, int publish, void *stream);

// halos and wrap columns of a whole buffer of `nrep` images with C channels from
// its interior (bulk mode, after a layer has been evaluated everywhere)
int ee_halo_bulk(const EeGeom *g, float *buf, int C, int nrep, void *stream);

// decoder: symbols of one step (packed [img][l]) + bias into ctx (nimg images).  flags != null
// (pinned host memory): the kernel first waits until flags[0] >= wait_for -- the host
// publishes the decoded symbols there -- so it can be queued before they exist; relay: a
// zeroed device int through which the one polling block tells the others.
int ee_scatter(const EeGeom *g, const float *packed, float *ctx, int lo, int len, int psum, float bias,
               int32_t *flags, int32_t *relay, int wait_for, void *stream);
// encoder: all symbols (NCHW float indices) + bias into a zeroed ctx
int ee_fill_ctx(const EeGeom *g, const float *symbols, float *ctx, float bias, void *stream);
// decoder epilogue: ctx -> NCHW symbols (index = value - bias), zero in dead columns
int ee_read_symbols(const EeGeom *g, const float *ctx, float *symbols, float bias, void *stream);
// integer CDF rows of one step from the last layer's output (unpadded, 3*ngroup
// channels); optional labels from the NCHW symbol tensor.  flags != null: once all rows are
// written the kernel stores flags[1] = publish (system scope); counter: a zeroed device int.
// packed != 0 (decoder form only: nstep 8, no symbols): 16-byte rows as above, one store per row
int ee_tables(const EeGeom *g, const float *y_last, const float *symbols, int32_t *table, int32_t *labels,
              int lo, int len, int psum, int nstep, float bias, float total, float beta, int32_t *counter,
              int32_t *flags, int publish, int packed, void *stream);
