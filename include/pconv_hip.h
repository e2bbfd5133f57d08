/*
 * pconv_hip.h -- C ABI of libpconv_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the reference's native module `PCONV`
 * (reference: extension/main.cpp:4-137).  The reference binds 21 C++ op classes
 * with pybind11; a maintainer replacing it binds the flat entry points below
 * instead (see INTEGRATION.md for the ctypes / pybind stub).  No torch types
 * cross this boundary: plain device pointers, dims and a hipStream_t.
 *
 * Conventions
 *   - every tensor is fp32, NCHW, contiguous; the tile-batch index is
 *     n*npart + tile (reference: sphere_slice_cuda.cu:98-99).
 *   - `stream` is a hipStream_t passed as void*; kernels are launched on it and
 *     never synchronise.
 *   - device functions return 0 on success, a negative PCONV_E* code otherwise;
 *     pconv_last_error() returns a thread-local message.  (The reference only
 *     printf()s on failure, caffe_cuda_macro.h:21-33; the Python shim raises.)
 *   - "host" functions touch no GPU state; they build the integer/float tables
 *     the kernels consume and are bit-exact restatements of the reference's
 *     one-off table kernels, with element offsets kept as integers (the
 *     reference stores them in fp32, pseudo_context_cuda.cu:97-99).
 */
#ifndef PCONV_HIP_H
#define PCONV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCONV_OK 0
#define PCONV_EINVAL (-1)  /* bad argument / shape            */
#define PCONV_ELAUNCH (-2) /* hip launch or runtime error     */
#define PCONV_ENOMEM (-3)

const char *pconv_last_error(void);
int pconv_abi_version(void);
/* number of devices visible to the HIP runtime (0 if none) */
int pconv_device_count(void);

/* ------------------------------------------------------------------------
 * Host-side geometry (replaces math_cuda.cu:177-253 and the one-off table
 * kernels of sphere_slice / sphere_uslice / pseudo_context / entropy_context)
 * ---------------------------------------------------------------------- */

/* Valid width of every latitude tile at tensor width `width`.
 * replaces sphere_cal_npart_hw_v3 (math_cuda.cu:223-253) and the width half of
 * sphere_cal_npart_hw_v2 (math_cuda.cu:177-221).  widths[npart]. */
int pconv_host_tile_widths(const float *weight, int npart, int height, int width,
                           int32_t *widths);

/* Catmull-Rom tap table of SphereSlice: for tile t, output column i < widths[t]
 * the first source column tap_col[t*width+i] and 4 coefficients.
 * replaces init_slice_param_kernel (sphere_slice_cuda.cu:13-32). */
int pconv_host_slice_taps(const int32_t *widths, int npart, int width,
                          int32_t *tap_col, float *tap_coef /* [npart*width*4] */);

/* Tap table of SphereUslice (up-resample from widths[t] to width).
 * replaces init_uslice_param_kernel (sphere_uslice_cuda.cu:13-30). */
int pconv_host_uslice_taps(const int32_t *widths, int npart, int width,
                           int32_t *tap_col, float *tap_coef);

/* Vertical-halo gather table of PseudoPad for tiles of `height` rows, `pad`
 * halo rows.  Entry e = ((t*2+side)*pad+r), side 0 = rows above, 1 = below.
 *   src_tile[e], src_row[e]      source tile / row inside that tile
 *   col[e*width+i], wgt[e*width+i]  first source column and its lerp weight
 * replaces pseudo_context_forward_kernel (pseudo_context_cuda.cu:51-104). */
int pconv_host_pad_table(const int32_t *widths, int npart, int height, int width, int pad,
                         int32_t *src_tile, int32_t *src_row, int32_t *col, float *wgt);

/* Wavefront schedule of the entropy model: positions (row*width+col) of the
 * npart stacked tiles sorted by plane row+col, valid columns only.
 * order[height*npart*width] (only plane_start[nplane] entries used),
 * plane_start[height*npart+width]  (nplane = height*npart+width-1, +1 end).
 * replaces entropy_context::reshape_hw (entropy_context_cuda.cu:13-45). */
int pconv_host_wavefront(const int32_t *widths, int npart, int height, int width,
                         int32_t *order, int32_t *plane_start);

/* Causal halo lists of the entropy model, one list per plane, deterministic
 * order.  Offsets are element offsets inside ONE image's channel-0 plane of the
 * padded tensor (npart, C, height+2*pad, width+2*pad), i.e. tile stride =
 * C*(height+2pad)*(width+2pad).
 *   entry k: dst[k], src0[k] (-1 = reads as zero), src1[k] (-2 = plain copy of
 *   src0), wgt[k], entry_plane[k];  plane_start[height*npart+width+pad] prefix
 *   offsets (nplane = height*npart+width+pad-1, +1 end).
 * Returns the number of entries, or <0.  Call with dst == NULL to size.
 * replaces entropy_context_kernel + step1/step2 + the CPU compaction
 * (entropy_context_cuda.cu:64-165,187-204). */
int pconv_host_causal_halo(const int32_t *widths, int npart, int channel, int height,
                           int width, int pad, int32_t *dst, int32_t *src0, int32_t *src1,
                           float *wgt, int32_t *entry_plane, int32_t *plane_start);

/* The same causal halo as a dense lookup table for kernels that compute halo
 * taps on the fly instead of reading stored ones: entry ((t*2+side)*pad+r)*width+i
 * = first source column (-1: only column 0 with weight 1-wgt; -2: no value, reads
 * as zero) and its weight. */
int pconv_host_causal_table(const int32_t *widths, int npart, int height, int width, int pad,
                            int32_t *col, float *wgt);

/* Viewport sampling table of MultiProject: tf[14*h_out*w_out*2] = (x, y) source
 * coordinates in an ERP of (height, width).
 * replaces projects_opt::init/update (projects_cuda.cu:7-165). */
int pconv_host_project_table(const float *theta, const float *phi, int nview, float fov,
                             int h_out, int w_out, int height, int width, float *tf);

/* ------------------------------------------------------------------------
 * Device kernels -- transform path
 * ---------------------------------------------------------------------- */

/* SphereSliceOp.forward  (sphere_slice_cuda.cu:87-146)
 * in (n, c, height, width) -> out (n*npart, c, height/npart + 2*pad, width + 2*pad);
 * only the interior is written when pad > 0, as in the reference. */
int pconv_sphere_slice(const float *in, float *out, const int32_t *widths,
                       const int32_t *tap_col, const float *tap_coef, int n, int c,
                       int height, int width, int npart, int pad, void *stream);

/* SphereUsliceOp.forward  (sphere_uslice_cuda.cu:73-126)
 * in (n*npart, c, h + 2*pad, width + 2*pad) -> out (n, c, h*npart, width) */
int pconv_sphere_uslice(const float *in, float *out, const int32_t *widths,
                        const int32_t *tap_col, const float *tap_coef, int n, int c, int h,
                        int width, int npart, int pad, void *stream);

/* PseudoPadOp.forward, the reference's three launches fused in one pass
 * (pseudo_pad.cu:39-125).  in (tn, c, h, w) -> out (tn, c, h+2p, w+2p). */
int pconv_pseudo_pad(const float *in, float *out, const int32_t *widths,
                     const int32_t *src_tile, const int32_t *src_row, const int32_t *col,
                     const float *wgt, int tn, int c, int h, int w, int pad, int npart,
                     void *stream);

/* The same when the producer has already written the tensor into the interior of
 * a padded buffer (pconv_conv2d / pconv_gdn with an output view): buf (tn, c,
 * h + 2*store, w + 2*store) holds the data at offset (store, store); only the ring
 * of the pad-p view (p <= store, origin (store - p, store - p)) is computed, in
 * place.  Tables as for pconv_pseudo_pad with the same pad. */
int pconv_pseudo_pad_ring(float *buf, const int32_t *widths, const int32_t *src_tile,
                          const int32_t *src_row, const int32_t *col, const float *wgt, int tn,
                          int c, int h, int w, int pad, int store, int npart, void *stream);

/* PseudoFillOp.forward, in place  (pseudo_fill_cuda.cu:28-62) */
int pconv_pseudo_fill(float *data, const int32_t *widths, int tn, int c, int h, int w,
                      int npart, int pad, int trim, float fvalue, void *stream);

/* DtowOp.forward  (dtow_cuda.cu:38-103); d2w != 0: (n,c,h,w)->(n,c/s^2,h*s,w*s) */
int pconv_dtow(const float *in, float *out, int n, int c, int h, int w, int stride, int d2w,
               void *stream);

/* PseudoQuantOp.forward (eval)  (pseudo_quant_cuda.cu:37-94,157-194)
 * weight (c, levels) raw parameter; level_tab (c, levels) scratch written here;
 * out_val / out_idx (idx as float, may be NULL); count (c, levels) histogram,
 * may be NULL. */
int pconv_quant(const float *x, const float *weight, float *level_tab, float *out_val,
                float *out_idx, float *count, const int32_t *widths, int tn, int c, int h,
                int w, int levels, int npart, void *stream);

/* ClipData.forward (model_zoo_v2.py:8-26), in place on n floats: x * 0.01 below 0, 1 + (x - 1) * 0.01 above 1,
 * x otherwise -- the same three roundings as the reference's masked assignments, in one pass. */
int pconv_leaky_clip(float *x, long long n, void *stream);

/* Frame I/O at the PCIe boundary: the device side of img2tensor / tensor2img (pseudo_codec.py:215-221).
 * u8_to_f32: in = n interleaved uint8 images (n, height, width, 3) ON THE DEVICE (copied there as they come from the
 * image file), out = float32 (n, 3, height, width) = float(u8) / 255.f, correctly rounded (the reference's numpy
 * float32 division).  f32_to_u8: out = (uint8)(int)(x * 255.f), numpy's float32 -> uint8 cast of tensor2img (truncation,
 * low byte kept).  width % 4 == 0; image 4-byte aligned, tensor 16-byte aligned. */
int pconv_frames_u8_to_f32(const uint8_t *in, float *out, int n, int height, int width, void *stream);
int pconv_frames_f32_to_u8(const float *in, uint8_t *out, int n, int height, int width, void *stream);

/* PseudoDQuantOp.forward  (pseudo_dquant_cuda.cu:24-70)
 * weight (wc, levels) raw parameter, level_tab (wc, levels) scratch */
int pconv_dquant(const float *x, const float *weight, float *level_tab, float *out,
                 const int32_t *widths, int tn, int c, int h, int w, int wc, int levels,
                 int npart, void *stream);

/* ProjectsOp.forward  (projects_cuda.cu:181-255)
 * in (n, c, height, width) -> out (nview*n, c, h_out, w_out), viewport-major */
int pconv_project(const float *in, const float *tf, float *out, int n, int c, int height,
                  int width, int nview, int h_out, int w_out, int nearest, void *stream);

/* ContextReshapeOp.forward (context_reshape_cuda.cu:30-60)
 * (n, g*cpg, h, w) -> (n*g*h*w, cpg) */
int pconv_context_reshape(const float *in, float *out, int n, int c, int h, int w, int ngroup,
                          void *stream);

/* ---- backward (transposed) forms of the linear geometry ops: training path, SURVEY 8f-4 ---- */

/* ContextReshapeOp.backward (context_reshape_cuda.cu:63-95): (n*g*h*w, cpg) -> (n, g*cpg, h, w) */
int pconv_context_reshape_backward(const float *top, float *bottom, int n, int c, int h, int w,
                                   int ngroup, void *stream);

/* SphereSliceOp.backward (sphere_slice_cuda.cu:191-244): grad of the tile stack
 * (n*npart, c, height/npart + 2*pad, width + 2*pad) -> grad of the image (n, c, height, width).
 * Same tap tables as the forward call. */
int pconv_sphere_slice_backward(const float *gout, float *gin, const int32_t *widths,
                                const int32_t *tap_col, const float *tap_coef, int n, int c,
                                int height, int width, int npart, int pad, void *stream);

/* SphereUsliceOp.backward (sphere_uslice_cuda.cu:128-200): grad of the image (n, c, h*npart, width)
 * -> grad of the tile stack (n*npart, c, h + 2*pad, width + 2*pad), zero outside the valid interior */
int pconv_sphere_uslice_backward(const float *gout, float *gin, const int32_t *widths,
                                 const int32_t *tap_col, const float *tap_coef, int n, int c, int h,
                                 int width, int npart, int pad, void *stream);

/* Reverse of pconv_host_pad_table: CSR over interior elements (tile*height + row)*width + col ->
 * halo entries that read the element, (destination tile << 24 | padded row*width + column, weight).
 * rev_start: npart*height*width + 1 ints; rev_dst / rev_wgt: up to 4*npart*pad*width records.
 * Returns the number of records.  Replaces pseudo_context_backward_kernel
 * (pseudo_context_cuda.cu:106-138), whose atomics-built lists have no defined order. */
int pconv_host_pad_reverse(const int32_t *widths, int npart, int height, int width, int pad,
                           int32_t *rev_start, int32_t *rev_dst, float *rev_wgt);

/* PseudoPadOp.backward (pseudo_pad.cu:127-235): grad (tn, c, h+2p, w+2p) -> (tn, c, h, w); gout is
 * not modified (the reference folds the wrap columns into its argument in place) */
int pconv_pseudo_pad_backward(const float *gout, float *gin, const int32_t *widths,
                              const int32_t *rev_start, const int32_t *rev_dst, const float *rev_wgt,
                              int tn, int c, int h, int w, int pad, int npart, void *stream);

/* PseudoEntropyPadOp.forward / backward (pseudo_entropy_pad_cuda.cu:39-241): the causal pad of the
 * training-time entropy net -- halo rows through pconv_host_causal_table, right wrap only, no pole
 * mirror.  in (tn, c, h, w) <-> out (tn, c, h+2p, w+2p). */
int pconv_entropy_pad(const float *in, float *out, const int32_t *widths, const int32_t *col,
                      const float *wgt, int tn, int c, int h, int w, int pad, int npart, void *stream);
int pconv_host_entropy_pad_table(const int32_t *widths, int npart, int height, int width, int pad,
                                 int version, int32_t *col, float *wgt);
int pconv_host_causal_reverse(const int32_t *widths, int npart, int height, int width, int pad,
                              int version, int32_t *rev_start, int32_t *rev_dst, float *rev_wgt);
int pconv_entropy_pad_backward(const float *gout, float *gin, const int32_t *widths,
                               const int32_t *rev_start, const int32_t *rev_dst, const float *rev_wgt,
                               int tn, int c, int h, int w, int pad, int npart, void *stream);

/* ProjectsOp.backward (projects_cuda.cu:257-329): gout (n*nview, c, h_out, w_out) -> gin and count
 * (n, c, height, width); count = the sampling weights each source pixel received. */
int pconv_project_backward(const float *gout, const float *tf, float *gin, float *count, int n, int c,
                           int height, int width, int nview, int h_out, int w_out, int nearest,
                           void *stream);

/* PseudoQuantOp.backward (pseudo_quant_cuda.cu:197-311).  x / val / idx: input and the two outputs
 * of the forward call, level_tab: the table that call filled; g_val / g_idx (may be NULL): gradients
 * of the outputs.  g_in (tn,c,h,w), g_weight (c,levels); bins: scratch of c*levels floats. */
int pconv_quant_backward(const float *x, const float *val, const float *idx, const float *g_val,
                         const float *g_idx, const float *level_tab, float *g_in, float *g_weight,
                         float *bins, const int32_t *widths, float top_alpha, int tn, int c, int h,
                         int w, int levels, int npart, void *stream);

/* MaskConstrainOp.forward, in place on a conv weight (nout, cin, k, k)
 * (mask_constrain_cuda.cu:19-88); constrain in {1,2,5,6} */
int pconv_mask_constrain(float *weight, int nout, int cin, int k, int ngroup, int constrain,
                         void *stream);

/* EntropyGmmOp.forward (entropy_gmm_cuda.cu:36-92): loss (m) plus the four
 * gradient buffers the reference keeps for backward (may be NULL). */
int pconv_gmm_loss(const float *weight, const float *delta, const float *mean,
                   const float *label, float *loss, float *d_weight, float *d_delta,
                   float *d_mean, float *d_label, int m, int ng, void *stream);

/* Dense per-tile convolution, implicit GEMM on fp32 MFMA (the nn.Conv2d call
 * sites of model_zoo_v2.py:41-45,83-86,100-105,119,143,158-164,181,205; cuDNN in
 * the reference).  in (tn, cin, h, w); packed_w from pconv_conv_pack_weight;
 * out (tn, cout, ho, wo), ho = (h - k)/stride + 1, no implicit padding.
 * k in {1,3}, stride in {1,2}.  act: 0 none, 1 PReLU(slope[cout]), 4 sigmoid.
 * col_limit (may be NULL): per latitude tile (t % npart) the first dead output
 * column; 64-column tiles starting at or beyond it are written as zeros without
 * being computed.
 * Every output is one k-ascending fp32 fmaf chain from 0 (k = (ci*k + kh)*k + kw),
 * then + bias, then the activation; then, each optional and in this order,
 * * gate[.], + residual[.] (both shaped like out, may be NULL: the attention
 * product and the residual sum that follow the convolution in
 * model_zoo_v2.py:53,76,93,114,175) and, with trim != 0, zero from col_limit on
 * (the PseudoFill that ends every block).
 * d2w != 0: DtowOp(2, d2w) (dtow_cuda.cu:38-75) applied by the store -- out is
 * (tn, cout/4, 2*ho, 2*wo); only with act 0/1 and no gate / residual / trim.
 * views (may be NULL = all dense NCHW): 12 element strides, (tile, channel, row)
 * for in, out, residual, gate in this order; columns are always contiguous.
 * Lets a convolution write the interior of a padded buffer (whose ring
 * pconv_pseudo_pad_ring then fills) and read such interiors. */
int pconv_conv_packed_size(int cout, int cin, int k, int *cout_pad, int *red_pad);
int pconv_conv_pack_weight(const float *w, float *packed, int cout, int cin, int k,
                           void *stream);
int pconv_conv2d(const float *in, const float *packed_w, const float *bias, float *out,
                 int tn, int cin, int h, int w, int cout, int k, int stride, int act,
                 const float *slope, const int32_t *col_limit, int npart,
                 const float *residual, const float *gate, int trim, int d2w,
                 const long long *views, void *stream);

/* The same convolution for k = 3, stride 1 by Winograd F(2x2, 3x3) on the fp32 matrix cores
 * (csrc/wino.hip): 4 instead of 9 multiply-adds per input channel, output channel and pixel -- the
 * algorithm cuDNN runs the reference's fp32 3x3 nn.Conv2d layers with (model_zoo_v2.py:41-45,83-86,
 * 158-164).  Results differ from pconv_conv2d's fmaf chain by rounding (~1e-6 relative).
 * packed_u from pconv_wino_pack_weight (pconv_wino_packed_size floats).  Arguments as pconv_conv2d
 * without k / stride / gate; act 0 or 1; views: 9 strides (in, out, residual) or NULL.  Takes the
 * layers pconv_wino_supported(...) accepts -- the single source of truth: cin % 16 == 0 (no ragged
 * channel path), cout >= 32, h, w >= 4, even output size, cout % 4 == 0 with d2w -- and returns
 * PCONV_EINVAL for anything else; rows of out / residual must start on 8-byte boundaries.  Input
 * views whose KC-channel chunk spans 4 GiB or more are refused (the LDS-DMA addresses a chunk as a
 * 64-bit uniform base + a 32-bit byte offset per lane); pconv_conv2d does the same for its 16-channel
 * chunks. */
long long pconv_wino_packed_size(int cout, int cin);
int pconv_wino_pack_weight(const float *w, float *packed, int cout, int cin, void *stream);
int pconv_wino_supported(int cin, int h, int w, int cout, int d2w);
int pconv_conv3x3_wino(const float *in, const float *packed_u, const float *bias, float *out,
                       int tn, int cin, int h, int w, int cout, int act, const float *slope,
                       const int32_t *col_limit, int npart, const float *residual, int trim,
                       int d2w, const long long *views, void *stream);
/* The same convolution with a 2-row x 128-column workgroup tile instead of 4 x 64 (csrc/wino_flat.hip): for launches of
 * TWO output rows (the remainders of PCONV.tile_conv2d's row split); same arguments, same packed weights, same bits per
 * output as pconv_conv3x3_wino. */
int pconv_conv3x3_wino_flat(const float *in, const float *packed_u, const float *bias, float *out,
                       int tn, int cin, int h, int w, int cout, int act, const float *slope,
                       const int32_t *col_limit, int npart, const float *residual, int trim,
                       int d2w, const long long *views, void *stream);

/* The same layers by Winograd F(4x2, 3x3) (csrc/wino42.hip): F(4, 3) vertically, F(2, 3) horizontally --
 * 3 instead of 4 (direct: 9) multiply-adds per input channel, output channel and pixel.  Same arguments and
 * epilogues as pconv_conv3x3_wino; takes the layers pconv_wino42_supported(...) accepts: cin % 24 == 0,
 * cout >= 32, at least 4 output rows, output width even (d2w: cout % 4 == 0).  Results differ from
 * pconv_conv2d's fmaf chain by rounding (~3e-6 relative; model_zoo_v2.py:41-45,83-86,158-164 are fp32
 * nn.Conv2d layers cuDNN runs with Winograd as well). */
long long pconv_wino42_packed_size(int cout, int cin);
int pconv_wino42_pack_weight(const float *w, float *packed, int cout, int cin, void *stream);
int pconv_wino42_supported(int cin, int h, int w, int cout, int d2w);
int pconv_conv3x3_wino42(const float *in, const float *packed_u, const float *bias, float *out,
                       int tn, int cin, int h, int w, int cout, int act, const float *slope,
                       const int32_t *col_limit, int npart, const float *residual, int trim,
                       int d2w, const long long *views, void *stream);

/* PseudoGDNV2.forward (PseudoContextV2.py:133-216) in one launch on the same
 * kernel: out = in / sqrt(beta + gamma * in^2) over channels (inverse: in * sqrt),
 * zeros from each tile's col_limit on (the reference's mask).  in, out
 * (tn, ch, h, w), distinct; packed_gamma = pconv_conv_pack_weight of the effective
 * (re-parametrised) gamma viewed as a (ch, ch, 1, 1) weight; beta (ch) effective. */
int pconv_gdn(const float *in, const float *packed_gamma, const float *beta, float *out,
              int tn, int ch, int h, int w, int inverse, const int32_t *col_limit, int npart,
              const float *residual /* added inside the valid columns, may be NULL */,
              const long long *views /* 9 strides: in, out, residual; may be NULL */,
              void *stream);

/* ------------------------------------------------------------------------
 * Device kernels -- entropy wavefront (one call = one step of one op)
 * `order`/`plane_start` come from pconv_host_wavefront; lo/hi = the range of
 * schedule entries handled by this step (plane_start[st] .. plane_start[end]).
 * ---------------------------------------------------------------------- */

/* DInput2Op.forward body (d_input_cuda_v2.cu:32-52): scatter packed symbols
 * (+bias) of the previous step into ctx (rep*nimg*npart, ngroup, h+2p, w+2p) */
int pconv_dinput2(const float *packed, float *ctx, const int32_t *order, int lo, int len,
                  int nimg, int ngroup, int npart, int h, int w, int pad, int psum,
                  float bias, int rep, void *stream);

/* EntropyCtxPadRun2Op.forward body, in place (entropy_ctx_pad_run2_cuda.cu:33-65).
 * list arrays from pconv_host_causal_halo; entry_plane[k] = plane of entry k. */
int pconv_ctx_pad_run2(float *data, const int32_t *dst, const int32_t *src0,
                       const int32_t *src1, const float *wgt, const int32_t *entry_plane,
                       int lo, int len, int nimg, int cpn, int channel, int npart, int h, int w,
                       int pad, int psum, void *stream);

/* EntropyConv2Op.forward{,_act,_batch,_act_batch} body
 * (entropy_conv_cuda_v2.cu:61-459).  x (nimg*npart, cin, h+2pi, w+2pi),
 * weight (nset, cout, cin, 5, 5), bias (nset, cout), slope NULL = no PReLU,
 * y (nimg*npart, cout, h+2po, w+2po) persistent.  nimg = images incl. replicas,
 * per_set = nimg / nset.  The step covers planes [first_plane, first_plane+nplane)
 * of the schedule (`plane_start` is the device copy of pconv_host_wavefront's
 * prefix array, max_plane_len the longest of those planes); the output group of
 * plane p is psum - p.
 * residual (may be NULL, same shape as y): added after the activation, i.e.
 * EntropyAddOp folded into the epilogue.
 * widths / vh_col / vh_wgt (may be NULL): pconv_host_causal_table of the input;
 * when given, halo taps are computed on the fly from tile interiors and the
 * stored halo of x is never read (no EntropyCtxPadRun2 needed). */
int pconv_entropy_conv(const float *x, const float *weight, const float *bias,
                       const float *slope, float *y, const int32_t *order,
                       const int32_t *plane_start, int first_plane, int nplane, int max_plane_len,
                       int nimg, int per_set, int cin, int cout, int ngroup, int k, int constrain,
                       int npart, int h, int w, int pad_in, int pad_out, int psum,
                       const float *residual, const int32_t *widths, const int32_t *vh_col,
                       const float *vh_wgt, void *stream);

/* EntropyAddOp.forward body, in place y += x (entropy_add_cuda.cu:25-44) */
int pconv_entropy_add(float *y, const float *x, const int32_t *order, int lo, int len,
                      int nimg, int channel, int ngroup, int npart, int h, int w, int pad,
                      int psum, void *stream);

/* DExtract2Op.forward body (d_extract_cuda_v2.cu:34-52): gather to packed list
 * out[(img*len + l)*cpn + ci] */
int pconv_dextract2(const float *x, float *out, const int32_t *order, int lo, int len,
                    int nimg, int channel, int cpn, int npart, int h, int w, int psum,
                    void *stream);

/* DExtract2Op.forward_batch body (d_extract_cuda_v2.cu:110-132): three packed
 * sections at distance `section_stride` */
int pconv_dextract2_batch(const float *x, float *out, const int32_t *order, int lo, int len,
                          int nimg, int channel, int cpn, int npart, int h, int w, int psum,
                          int nout, long long section_stride, void *stream);

/* EntropyGmmTableOp.forward_batch / forward (entropy_gmm_table_cuda.cu:29-185).
 * weight/delta/mean are (tn, ng) each, modified in place like the reference
 * (softmax / relu+beta); table (tn, nstep+1) holds integers as floats.
 * batch_arith != 0 selects forward_batch's mixed float/double accumulation
 * (:136-153), 0 the all-float accumulation of forward (:59-80). */
int pconv_gmm_table(float *weight, float *delta, const float *mean, float *table, int tn,
                    int ng, int nstep, float bias, float total, float beta, int batch_arith,
                    void *stream);

/* Engine step: DExtract2Batch + EntropyBatchGmmTable (+ the label DExtract2) in
 * one launch, integer output.  y (3*nimg*npart, 3*ngroup, h, w) = last layer;
 * table int32 [nimg*len][nstep+1]; symbols (nimg*npart, ngroup, h, w) / labels
 * int32 [nimg*len] may both be NULL. */
int pconv_step_tables(const float *y, const float *symbols, int32_t *table, int32_t *labels,
                      const int32_t *order, int lo, int len, int nimg, int ngroup, int npart, int h,
                      int w, int psum, int nstep, float bias, float total, float beta, void *stream);

/* encoder side of DInput2: scatter ALL symbols at once, ctx (rep*tn, c, h+2p, w+2p)
 * interior = symbol + bias inside the valid width (the buffer must be zeroed) */
int pconv_symbols_to_ctx(const float *symbols, float *ctx, const int32_t *widths, int tn, int c,
                         int h, int w, int pad, int npart, float bias, int rep, void *stream);

/* decoded symbols out of the padded context tensor (tn, c, h+2p, w+2p):
 * out (tn, c, h, w) = interior + bias inside each tile's valid width, 0 elsewhere
 * (pseudo_codec.py:159-160) */
int pconv_ctx_to_symbols(const float *ctx, float *out, const int32_t *widths, int tn, int c, int h,
                         int w, int pad, int npart, float bias, void *stream);

/* ------------------------------------------------------------------------
 * Native entropy engine: the EntEncoder / EntDecoder loops
 * (pseudo_codec.py:97-114,145-160) as one C++ host loop over the step kernels
 * above, `nimg` frames in lock-step, one arithmetic-coded stream per frame.
 * The streams are byte-identical to what the per-op path writes.
 * ---------------------------------------------------------------------- */
typedef struct pconv_entropy_engine pconv_entropy_engine;

/* h, w: rows per tile and columns of the symbol tensor (after Dtow);
 * tile_weight[npart] as for pconv_host_tile_widths; bias = (levels-1)/2. */
pconv_entropy_engine *pconv_ee_create(int npart, int ngroup, int h, int w, int nimg,
                                      const float *tile_weight, float bias, int nlevels, float total,
                                      float beta);
void pconv_ee_destroy(pconv_entropy_engine *e);
/* A non-blocking HIP stream on the current device, created directly (not out of a framework's stream pool): the
 * copy streams of the frame pipe (host <-> HBM beside the compute).  The caller destroys it. */
int pconv_stream_create(void **stream);
int pconv_stream_destroy(void *stream);
/* layer 0..11 = net.0.conv, net.1.conv1.conv, net.1.conv2.conv, ..., net.6.conv;
 * device pointers: weight (3, 3G, cin, 5, 5), bias (3, 3G), slope (3, 3G) or NULL.
 * The weight is copied (re-packed) on `stream`; bias and slope are borrowed. */
int pconv_ee_set_layer(pconv_entropy_engine *e, int layer, const float *weight, const float *bias,
                       const float *slope, void *stream);
long long pconv_ee_symbols_per_image(const pconv_entropy_engine *e);
int pconv_ee_steps(const pconv_entropy_engine *e);
/* Host side of the engine (no reference counterpart: the reference drives its loop from one Python
 * thread, pseudo_codec.py:145-160; its only multi-process code is one process per GPU,
 * test/trainDDP_Full.py:83-86,201-204).  pconv_ee_host_cpus: CPUs this rank's host threads can count
 * on = min(affinity mask, cgroup CPU quota / LOCAL_WORLD_SIZE).  pconv_ee_spin_us: microseconds an
 * idle decode worker polls before it blocks when a call runs `call_threads` host threads (one per
 * frame): 2000 (through the GPU part of a step) only if call_threads + 1 <= pconv_ee_host_cpus(),
 * else 60; PCONV_ENGINE_SPIN_US overrides.  Neither touches the GPU. */
int pconv_ee_host_cpus(void);
int pconv_ee_spin_us(int call_threads);
/* How an engine of `nimg` frames lays out its host side here (any pointer may be NULL): lock-step groups (= decoder
 * chains = driver threads), threads that arithmetic-decode a group's frames (0 = one per frame), queued-ahead (1)
 * or host-driven (0) chain, waits that sleep instead of spinning.  With nimg + 1 <= pconv_ee_host_cpus(): the
 * measured best of profiles/round3_decode_groups.txt; otherwise (a rank with a small share of the host) at most
 * one group per CPU, one decoding thread per group, host-driven chain, sleeping waits (blocking events)
 * (profiles/round5_host_share.txt).  PCONV_ENGINE_GROUPS / _WORKERS / _CHAIN / _BLOCKING_SYNC override. */
int pconv_ee_host_plan(int nimg, int *groups, int *group_threads, int *queued_chain, int *blocking_sync);
/* How the host threads of THIS engine wait for the GPU: 0 = the runtime's default (spinning) stream waits, 1 = sleeping
 * waits on blocking events (hipEventBlockingSync; the plan's blocking_sync, fixed at pconv_ee_create).  No device-wide
 * schedule flag is set either way. */
int pconv_ee_wait_mode(const pconv_entropy_engine *e);
/* Explicit, process-wide opt-in: hipSetDeviceFlags(hipDeviceScheduleBlockingSync) on the current device (enable != 0),
 * then the flag is read back: returns 1 when every runtime wait on the device now sleeps, 0 when it does not (not asked
 * for, or refused by the runtime on a live context -- the engine's blocking events still apply then).  Never called by
 * the library itself. */
int pconv_device_blocking_sync(int enable);
/* symbols: device float (nimg*npart, ngroup, h, w), dead columns zero */
int pconv_ee_encode(pconv_entropy_engine *e, const float *symbols, void *stream);
/* the same in two halves: begin queues the GPU part IN `stream` and starts the host thread that
 * arithmetic-codes the frames as their tables arrive, then returns; end joins that thread.
 * `symbols` must stay alive until end.  Between the two the caller may queue other GPU work in
 * `stream` (the transforms of the next frames): it runs while the CPU codes. */
int pconv_ee_encode_begin(pconv_entropy_engine *e, const float *symbols, void *stream);
/* The last group of an encode call is evaluated in `nrange` wavefront step ranges (1 .. 8; 0 = the default: 4, or
 * PCONV_ENGINE_ENCODE_RANGES), each range's CDF rows copied to the host as soon as they exist, so that the
 * arithmetic coder starts before the GPU has finished the group.  A caller that pipelines several encode calls
 * (engine.CodecEngine.encode: chunk k is coded on the CPU under chunk k + 1's GPU work) asks for 1 on all but the
 * last: ranges re-evaluate the blocks on their boundaries.  Same streams for every value. */
int pconv_ee_set_encode_ranges(pconv_entropy_engine *e, int nrange);
int pconv_ee_encode_end(pconv_entropy_engine *e, void *stream);
const uint8_t *pconv_ee_stream(const pconv_entropy_engine *e, int img, size_t *nbytes);
int pconv_ee_decode(pconv_entropy_engine *e, const uint8_t *const *streams, const size_t *nbytes,
                    float *symbols_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCONV_HIP_H */
