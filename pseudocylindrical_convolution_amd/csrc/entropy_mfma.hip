// Encoder ("bulk") form of a layer of the entropy network on the fp32 matrix cores.
//
// Same numbers as ee_conv_bulk_kernel / ee_step_kernel (entropy_engine.hip), bit for bit: the
// published order of the masked 5 x 5 convolution (entropy.hip; the reference:
// entropy_conv_cuda_v2.cu:326-380, driven by pseudo_codec.py:97-114) is
//   lane l of 64 sums the reduction entries kk = l, l + 64, l + 128, ... as ONE fmaf chain
//   from 0, then the 64 partial sums meet in an xor butterfly 32, 16, 8, 4, 2, 1.
// A v_mfma_f32_16x16x4_f32 IS a k-ascending fmaf chain (tools/mfma16_chain_probe.hip: 0 of
// 256 outputs differ), so the chain of "lane class" l is a GEMM of its own,
//   C_l[out][pos] = sum_j W[out][l + 64 j] * X[pos][l + 64 j],  j = 0 .. ITER - 1 ascending,
// 17 deep for 42 input channels (5 instructions, K padded to 20 with zero weights: a chain
// that never holds -0 is unchanged by + (+-0)), and the butterfly is the fixed tree
//   s1[l] = C_l + C_(l^32), s2[l] = s1[l] + s1[l^16], ... , total = s5[0] + s5[1]
// (float addition commutes, so every lane of the butterfly holds this one value).  The classes
// are visited in bit-reversed order, which makes the tree a stack of at most six partial sums:
// class order index i = 8 a + b  ->  l = 8 bitrev3(b) + bitrev3(a); after class i as many
// levels fold as i has trailing ones.  Masked taps are zeros in the packed weights, as in
// the slabs of the other kernels: on the matrix cores they cost nothing extra.
//
// Work split.  A workgroup takes a block of BR rows x BC columns of one tile of one replica
// (set x image) and stages its (BR + 4) x (BC + 4) x CIN patch of the channels-last padded
// input in LDS ONCE, by LDS-DMA (the vector kernel gathers every window from L2: 10.85 M L2
// requests per launch, "waits on its window gathers", profiles/round4_bench_pmc.json).  A wave
// owns kNT stacked rows x 16 columns = kNT position tiles (matrix columns) and all 48 (42)
// outputs = three output tiles (matrix rows).  The weights come pre-packed as MFMA
// A-fragments in class order (ee_pack_weight_mfma: 4 KB per class and set).  Forms (measured:
// profiles/round5_entropy_mfma_variants.txt; PCONV_EE_MFMA_NT / _WAVES / _WSRC):
//   default   one row per wave, four waves: 15 MFMAs per class and wave, 144 registers (three
//             waves per SIMD); fragments straight from global memory into registers and the
//             patch entries from LDS, both for class i + 1 under class i's MFMAs; no barrier
//             after the prologue.  Also takes the 14-channel input layer.
//   ring      the same with the fragments through a 4-slot LDS ring (register-staged two
//             classes ahead, one barrier per class): level with the default.
//   two rows  30 MFMAs per class and wave (every fragment feeds two), 250 registers, two waves
//             per SIMD, ring only: 5-10 % slower.
//
// The causal masks make a (position, group) output independent of everything the decoder
// would not have yet, so -- like ee_conv_bulk_kernel -- the kernel may evaluate ALL groups of
// a position at once; a step range [s_lo, s_hi) only filters the stores (and skips blocks
// that hold no pair of the range).
#include <stdlib.h>
#include <atomic>
#include "common.h"
#include "ee_kernels.h"

namespace {

constexpr int kWave = 64;
constexpr int K5 = 5, KK = 25, HALF = 2, PAD = 2, GO = 3;
constexpr int kMT = 3;     // output tiles of 16: 42 -> 48 matrix rows
constexpr int kRing = 4;   // LDS ring slots of one class each
constexpr int kPatchSlack = 256;  // floats behind the patch that entries past the reduction length may read
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int bitrev3(int v) { return ((v & 1) << 2) | (v & 2) | ((v >> 2) & 1); }
__host__ __device__ constexpr int iter_of(int cin) { return (cin * KK + kWave - 1) / kWave; }
__host__ __device__ constexpr int steps_of(int cin) { return (iter_of(cin) + 3) / 4; }          // MFMAs (K = 4) per chain
__host__ __device__ constexpr int quads_of(int cin) { return (steps_of(cin) * kMT + 3) / 4; }   // 16-byte pieces per lane and class
__host__ __device__ constexpr int frag_floats(int cin) { return quads_of(cin) * kWave * 4; }

// weights (nset, cout, cin, 5, 5) -> [set][class order index i][quad][lane][4]: element e of quad q is
// slot s = 4 q + e = 3 m + mt, the A operand of MFMA step m for output tile mt:
//   lane L holds W[out = 16 mt + (L & 15)][kk = l(i) + 64 (4 m + (L >> 4))]
// with the causal mask of the output's group applied (pack_weight_kernel's rule), zero past the
// reduction length, past the last output and in the padding slots.
__global__ void pack_weight_mfma_kernel(const float *__restrict__ w, float *__restrict__ packed, int cin, int cout,
                                        int ngroup, int slack, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int quads = quads_of(cin), steps = steps_of(cin), red = cin * KK;
  const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
  long long r = i >> 8;
  const int quad = (int)(r % quads);
  r /= quads;
  const int ci_ = (int)(r & 63), set = (int)(r >> 6);
  const int slot = quad * 4 + e;
  float v = 0.f;
  if (slot < steps * kMT) {
    const int m = slot / kMT, mt = slot - m * kMT;
    const int a = ci_ >> 3, b = ci_ & 7;
    const int l = bitrev3(b) * 8 + bitrev3(a);
    const int kk = l + kWave * (4 * m + (lane >> 4));
    const int out = 16 * mt + (lane & 15);
    if (kk < red && out < cout) {
      const int tc = out / GO, o = out - tc * GO, group_in = cin / ngroup;
      const int tap = kk / cin, ci = kk - tap * cin;
      const int kh = tap / K5, kw = tap - kh * K5;
      const bool ok = (2 * HALF - kh - kw) * group_in - ci + (tc + slack) * group_in > 0;
      if (ok) v = w[(((size_t)set * ngroup + tc) * GO + o) * red + ci * KK + tap];
    }
  }
  packed[i] = v;
}

// kNT: position tiles of a wave = stacked rows x 16 columns.  2: every weight fragment feeds two MFMAs, 250
// registers, two waves per SIMD; 1: 15 MFMAs per class and wave, <= 168 registers, three waves per SIMD
// DIRECT (with kNT == 1): no LDS ring and no barrier in the class loop -- every wave fetches the next class's
// fragments itself, straight from global memory into registers (4 KB per wave and class, the four waves of a
// workgroup and its neighbours on the CU read the same lines within a few classes of each other: L1 / L2 hits),
// under the current class's matrix instructions.
template <int CIN, int WAVES, int kNT, bool DIRECT>
__global__ __launch_bounds__(WAVES * kWave, kNT == 1 ? 3 : 2) void ee_conv_bulk_mfma_kernel(
    EeGeom g, const int4 *__restrict__ blocks, int rp_n, int ct_n, const float *__restrict__ x, int shared_input,
    const float *__restrict__ wfrag, const float *__restrict__ bias, const float *__restrict__ slope,
    const float *__restrict__ residual, float *__restrict__ y, int pad_out, int s_lo, int s_hi) {
  constexpr int STEPS = steps_of(CIN), QUADS = quads_of(CIN), FRAG = frag_floats(CIN);
  constexpr int COUT = 3 * (CIN == 14 ? 14 : CIN / 3);  // 3 per group
  static_assert(COUT <= 16 * kMT, "three output tiles");
  static_assert(QUADS <= 4, "a class's fragments are four 16-byte pieces per lane at most");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(!DIRECT || kNT == 1, "direct weight fetch goes with the pipelined one-row form");
  float *ring = smem;                                  // kRing x FRAG floats (none with DIRECT)
  float *patch = smem + (DIRECT ? 0 : kRing * FRAG);   // (BR + 4) x (BC + 4) x CIN
  typedef const __attribute__((address_space(4))) int32_t const_i32_t;
  const_i32_t *brec = (const_i32_t *)(blocks + blockIdx.x);
  const int tile = brec[0], row0 = brec[1], col0 = brec[2];
  const int pn = blockIdx.y;  // replica-major image index: set * nimg + img
  const int set = pn / g.nimg;
  const int h = g.h, w = g.w;
  const int BR = kNT * rp_n, BC = 16 * ct_n, PW = BC + 4, PR = BR + 4;
  const int width = ((const_i32_t *)g.widths)[tile];
  {
    // any (position, group) pair of the step range in this block?  (uniform: before any barrier)
    const int cmax = (col0 + BC < width ? col0 + BC : width) - 1;
    const int pmin = tile * h + row0 + col0, pmax = tile * h + row0 + BR - 1 + cmax;
    if (pmax + g.ngroup - 1 < s_lo || pmin >= s_hi) return;
  }
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  // Weight fragments of class i: 4 KB, one piece per thread (16 bytes with four waves, 8 with eight), staged
  // through registers (global load at step i - 2, LDS store at step i - 1, first read behind the barrier of step
  // i).  Not LDS-DMA: the compiler puts a vmcnt(0) in front of every LDS read that follows a DMA it cannot prove
  // disjoint, i.e. it would wait for the piece it has just requested in every class.  And EVERY thread moves a
  // piece: behind a guarded load the compiler's wait-count state of the skipped path made it wait for the load
  // it had just issued.
  const float *wset = wfrag + (size_t)set * 64 * FRAG;
  constexpr int PIECE = DIRECT ? 4 : FRAG / (WAVES * kWave);  // floats per thread and class (ring form only)
  static_assert(DIRECT || (PIECE * WAVES * kWave == FRAG && (PIECE == 4 || PIECE == 2)), "whole pieces");
  typedef float piece_t __attribute__((ext_vector_type(PIECE)));
  auto fetch_class = [&](int i) { return *reinterpret_cast<const piece_t *>(wset + (size_t)i * FRAG + tid * PIECE); };
  auto store_class = [&](int i, const piece_t &v) {
    *reinterpret_cast<piece_t *>(ring + (i & (kRing - 1)) * FRAG + tid * PIECE) = v;
  };
  // PIPE (one row per wave: registers to spare): the operands of class i + 1 are read from LDS while the matrix
  // instructions of class i run, so the weight ring runs one class further ahead
  constexpr bool PIPE = kNT == 1;
  piece_t wnext = {};
  if (!DIRECT) {
    wnext = fetch_class(0);
    store_class(0, wnext);
    if (PIPE) store_class(1, fetch_class(1));
    wnext = fetch_class(PIPE ? 2 : 1);
  }
  {
    // the patch: PR rows of PW * CIN contiguous floats each (a pixel is 168 bytes), by LDS-DMA in 16-byte pieces
    // (8-byte aligned sources: tools/dma16_probe.hip), every piece of the workgroup in flight at once -- a loop of
    // load / store pairs was a chain of 24 memory round trips, as long as the 64 classes of matrix work.  LDS
    // piece p = patch row p / row16, piece p % row16 of that row; the tail of the last round re-reads the last
    // piece into the padding behind the patch.  Columns past the padded row of the buffer (blocks at the right
    // edge of a full-width tile) repeat its last piece: only dead positions read them.
    typedef __attribute__((address_space(3))) void lds_ptr_t;
    typedef const __attribute__((address_space(1))) void glb_ptr_t;
    const int xi = shared_input ? pn % g.nimg : pn;
    const size_t tile_elems = (size_t)(h + 2 * PAD) * (w + 2 * PAD) * CIN;
    const float *xt = x + ((size_t)xi * g.npart + tile) * tile_elems + ((size_t)row0 * (w + 2 * PAD) + col0) * CIN;
    const int row16 = PW * CIN / 4;                       // (PW is even: whole pieces)
    const int lim16 = (w + 2 * PAD - col0) * CIN / 4;     // whole pieces left in the buffer's row
    const int npiece = PR * row16;
    const size_t buf_pitch = (size_t)(w + 2 * PAD) * CIN;
    // (piece p = p0 + tid: its patch row and place in the row advance with the round, no division per piece)
    int pr = tid / row16, j = tid - pr * row16;
    const int dpr = (WAVES * kWave) / row16, dj = (WAVES * kWave) - dpr * row16;
    for (int p0 = 0; p0 < npiece; p0 += WAVES * kWave) {
      const bool in = p0 + tid < npiece;
      const int prc = in ? pr : PR - 1, jc = in ? j : row16 - 1;  // (the tail re-reads the last piece)
      const float *src = xt + prc * buf_pitch + 4 * (jc < lim16 ? jc : lim16 - 1);
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)src, (lds_ptr_t *)(patch + (size_t)(p0 + wave * kWave) * 4), 16, 0, 0);
      pr += dpr;
      j += dj;
      if (j >= row16) {
        j -= row16;
        pr++;
      }
    }
    // the slack behind the (rounded) patch that over-long entries of the last rows may read
    const int rounded = (npiece + WAVES * kWave - 1) / (WAVES * kWave) * (WAVES * kWave) * 4;
    for (int k = tid; k < kPatchSlack; k += WAVES * kWave) patch[rounded + k] = 0.f;
  }
  // this wave's positions: rows row0 + kNT rp + {0 .. kNT - 1}, columns col0 + 16 ct + (lane & 15)
  const int rp = wave % rp_n, ct = wave / rp_n;
  const int q = lane >> 4;
  const bool live = col0 + 16 * ct < width;  // (wave-uniform) any live column in its tiles
  const int lane_base = ((kNT * rp) * PW + 16 * ct + (lane & 15)) * CIN;  // window origin of tile 0, in floats
  const int row_pitch = PW * CIN;
  const int kh_stride = (PW - K5) * CIN;
  // Byte offset of reduction entry kk = l + 64 (4 m + q) of this lane's window: 4 (lane_base + kk + kh * kh_stride),
  // kh = kk / (5 CIN) the window row.  While the class l runs over 0 .. 63 an entry's row changes at most once
  // (64 < 5 CIN), at l = thr[m] (64: never): off = base[m] + 4 l (+ 4 kh_stride from thr[m] on) -- a compare, a
  // select and an add per MFMA step instead of a division by 210.  Entries past the reduction length (zero
  // weights: any finite value will do) stay in window row 4 and run at most 230 floats past the window's end:
  // into the patch's following rows or the zeroed kPatchSlack behind it.
  unsigned base_b[STEPS];
  int thr[STEPS];
#pragma unroll
  for (int m = 0; m < STEPS; m++) {
    const int c = kWave * (4 * m + q);
    int kh0 = c / (K5 * CIN);
    kh0 = kh0 < K5 - 1 ? kh0 : K5 - 1;
    const int cross = (kh0 + 1) * K5 * CIN - c;  // first l in the next window row
    thr[m] = (kh0 < K5 - 1 && cross < kWave) ? cross : kWave;
    base_b[m] = 4u * (unsigned)(lane_base + c + kh0 * kh_stride);
  }

  f32x4 S1[kMT * kNT], S2[kMT * kNT], S3[kMT * kNT], P3[kMT * kNT], P4[kMT * kNT], P5[kMT * kNT];
#pragma unroll
  for (int t = 0; t < kMT * kNT; t++) S1[t] = S2[t] = S3[t] = P3[t] = P4[t] = P5[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // operands of a class: A fragments (slot s = 3 m + mt of the ring slot) and this lane's B entries of the patch
  // (slot: ring slot of the class; with DIRECT: the class order index, fragments straight from global memory)
  auto read_operands = [&](int slot, int l, float (&af)[QUADS * 4], float (&bf)[STEPS][kNT]) {
    const float4 *fr = reinterpret_cast<const float4 *>(DIRECT ? wset + (size_t)slot * FRAG : ring + slot * FRAG) + lane;
#pragma unroll
    for (int qd = 0; qd < QUADS; qd++) {
      const float4 v = fr[qd * kWave];
      af[4 * qd] = v.x, af[4 * qd + 1] = v.y, af[4 * qd + 2] = v.z, af[4 * qd + 3] = v.w;
    }
#pragma unroll
    for (int m = 0; m < STEPS; m++) {
      const unsigned off = base_b[m] + 4u * (unsigned)l + (l >= thr[m] ? 4u * (unsigned)kh_stride : 0u);
#pragma unroll
      for (int nt = 0; nt < kNT; nt++)
        bf[m][nt] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(patch) + off + nt * 4 * row_pitch);
    }
  };
  float af[QUADS * 4], bf[STEPS][kNT];
  if (PIPE) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the patch DMA, classes 0 and 1
    if (DIRECT && !live) return;  // (no barrier from here on)
    if (live) read_operands(0, 0, af, bf);
  }
#pragma unroll 1
  for (int a = 0; a < 8; a++) {
    const int la = bitrev3(a);
    f32x4 Q[kMT * kNT];
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const int i = a * 8 + b;
      float afn[QUADS * 4], bfn[STEPS][kNT];
      if (PIPE) {
        if (!DIRECT) {
          // class i + 1 is in its slot (stored during step i - 1); the barrier waits for the LDS stores only --
          // __syncthreads() would also wait for the weight load in flight
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          store_class(i + 2, wnext);  // (slot (i + 2) & 3 was last read in step i - 3; past the end: a dummy)
          wnext = fetch_class(i + 3 < 64 ? i + 3 : 63);
          if (!live) continue;  // (wave-uniform; the barriers stay outside)
        }
        // the next class's operands, in flight under this class's matrix instructions
        const int ln = b < 7 ? bitrev3(b + 1) * 8 + la : bitrev3((a + 1) & 7);
        read_operands(DIRECT ? (i + 1 < 64 ? i + 1 : 63) : (b + 1) & (kRing - 1), ln, afn, bfn);  // (behind the last class: a dummy)
      } else {
        // class i is in its slot (stored during step i - 1; at i = 0: with the patch)
        if (a == 0 && b == 0)
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the patch DMA of every wave)
        else
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        store_class(i + 1, wnext);  // (slot (i + 1) & 3 was last read in step i - 3; past the end: a dummy)
        wnext = fetch_class(i + 2 < 64 ? i + 2 : 63);
        if (!live) continue;
        read_operands(b & (kRing - 1), bitrev3(b) * 8 + la, af, bf);
      }
      f32x4 acc[kMT * kNT];
#pragma unroll
      for (int t = 0; t < kMT * kNT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < STEPS; m++)
#pragma unroll
        for (int mt = 0; mt < kMT; mt++)
#pragma unroll
          for (int nt = 0; nt < kNT; nt++)
            acc[mt * kNT + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m * kMT + mt], bf[m][nt], acc[mt * kNT + nt], 0, 0, 0);
      if (PIPE) {
#pragma unroll
        for (int k = 0; k < QUADS * 4; k++) af[k] = afn[k];
#pragma unroll
        for (int m = 0; m < STEPS; m++)
#pragma unroll
          for (int nt = 0; nt < kNT; nt++) bf[m][nt] = bfn[m][nt];
      }
      // the butterfly's tree, depth first: as many levels fold as b has trailing ones
#pragma unroll
      for (int t = 0; t < kMT * kNT; t++) {
        if ((b & 1) == 0) {
          S1[t] = acc[t];
        } else {
          const f32x4 u1 = S1[t] + acc[t];  // xor 32
          if ((b & 2) == 0) {
            S2[t] = u1;
          } else {
            const f32x4 u2 = S2[t] + u1;  // xor 16
            if ((b & 4) == 0)
              S3[t] = u2;
            else
              Q[t] = S3[t] + u2;  // xor 8
          }
        }
      }
    }
    if (live) {
      // levels xor 4, 2, 1 over the outer index (uniform branches)
      if ((a & 1) == 0) {
#pragma unroll
        for (int t = 0; t < kMT * kNT; t++) P3[t] = Q[t];
      } else if ((a & 2) == 0) {
#pragma unroll
        for (int t = 0; t < kMT * kNT; t++) P4[t] = P3[t] + Q[t];
      } else if ((a & 4) == 0) {
#pragma unroll
        for (int t = 0; t < kMT * kNT; t++) P5[t] = P4[t] + (P3[t] + Q[t]);
      } else {  // a == 7: the totals (kept in P3)
#pragma unroll
        for (int t = 0; t < kMT * kNT; t++) P3[t] = P5[t] + (P4[t] + (P3[t] + Q[t]));
      }
    }
  }
  if (!live) return;
  // way out: lane L holds outputs 16 mt + 4 (L >> 4) + r of position (row 2 rp + nt, column L & 15)
  const int col = col0 + 16 * ct + (lane & 15);
  if (col >= width) return;
  // four consecutive outputs per accumulator tile: 8-byte pieces (a pixel is 168 bytes, an output quad starts at a
  // multiple of 16: both 8-byte aligned); the whole schedule (no step range) skips the per-output range test
  const bool whole = s_lo <= 0 && s_hi >= g.h * g.npart + g.w + g.ngroup - 2;  // (uniform; the schedule has rows + w + ngroup - 2 steps)
  typedef float f2 __attribute__((ext_vector_type(2)));
  const float *bset = bias + set * COUT, *sset = slope ? slope + set * COUT : nullptr;
#pragma unroll
  for (int mt = 0; mt < kMT; mt++) {
    const int out0 = 16 * mt + 4 * q;
    if (out0 >= COUT) continue;  // (the last quads of the padded tile: lanes q = 3 of tile 2, and q = 2's second half below)
    const bool second = out0 + 2 < COUT;
    const f2 b01 = *reinterpret_cast<const f2 *>(bset + out0);
    const f2 b23 = second ? *reinterpret_cast<const f2 *>(bset + out0 + 2) : (f2){0.f, 0.f};
    f2 s01 = {1.f, 1.f}, s23 = {1.f, 1.f};  // (v * 1 is v)
    if (sset) {
      s01 = *reinterpret_cast<const f2 *>(sset + out0);
      if (second) s23 = *reinterpret_cast<const f2 *>(sset + out0 + 2);
    }
#pragma unroll
    for (int nt = 0; nt < kNT; nt++) {
      const int row = row0 + kNT * rp + nt;
      const int plane = tile * h + row + col;
      const size_t ob = ((((size_t)pn * g.npart + tile) * (h + 2 * pad_out) + row + pad_out) * (w + 2 * pad_out) + col + pad_out) * COUT + out0;
      f2 r01 = {0.f, 0.f}, r23 = {0.f, 0.f};
      if (residual) {
        r01 = *reinterpret_cast<const f2 *>(residual + ob);
        if (second) r23 = *reinterpret_cast<const f2 *>(residual + ob + 2);
      }
      const f32x4 t = P3[mt * kNT + nt];
      float v[4] = {t[0] + b01[0], t[1] + b01[1], t[2] + b23[0], t[3] + b23[1]};
      const float sl[4] = {s01[0], s01[1], s23[0], s23[1]}, rs[4] = {r01[0], r01[1], r23[0], r23[1]};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (v[r] < 0) v[r] = v[r] * sl[r];
        if (residual) v[r] = v[r] + rs[r];
      }
      if (whole) {
        *reinterpret_cast<f2 *>(y + ob) = (f2){v[0], v[1]};
        if (second) *reinterpret_cast<f2 *>(y + ob + 2) = (f2){v[2], v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int step = plane + (out0 + r) / GO;
          if (out0 + r < COUT && step >= s_lo && step < s_hi) y[ob + r] = v[r];
        }
      }
    }
  }
}


// ---- four lane classes per instruction: v_mfma_f32_16x16x1_4b_f32 -------------------------------------------
//
// Four independent 16 x 16 x 1 blocks per instruction (tools/mfma16x1_4b_probe.hip: every block an exact fmaf
// step, 32 cycles like the 16 x 16 x 4 form): block b carries lane class l0 + {0, 32, 16, 48}[b], the K = 17 chain of
// a class is 17 instructions -- no padding to 20 -- and the first two levels of the butterfly,
// (C_l0 + C_l0+32) + (C_l0+16 + C_l0+48), are sums of a lane's own four result blocks.  The sixteen groups l0 are
// visited in bit-reversed order (group index 4 a + b -> l0 = 4 bitrev2(b) + bitrev2(a)), the levels xor 8, 4, 2, 1
// are a stack of four.  Operands: lane L = block L >> 4, row / column L & 15: A = W[out 16 mt + (L & 15)][kk],
// B = X[pos L & 15][kk], kk = l0 + dcls(L >> 4) + 64 j for step j.  A group's 51 fragments are 13 16-byte pieces
// per lane (ee_pack_weight_mfma4), fetched straight from global memory: the piece of the NEXT group replaces a
// piece as soon as its last step has issued, likewise the patch entry of step j -- one whole group (1 632 matrix
// cycles) of distance without a second register set.
constexpr int kDcls[4] = {0, 32, 16, 48};
__host__ __device__ constexpr int bitrev2(int v) { return ((v & 1) << 1) | ((v >> 1) & 1); }
__host__ __device__ constexpr int slots4_of(int cin) { return iter_of(cin) * kMT; }           // 51
__host__ __device__ constexpr int quads4_of(int cin) { return (slots4_of(cin) + 3) / 4; }     // 13
__host__ __device__ constexpr int frag4_floats(int cin) { return quads4_of(cin) * kWave * 4; }

// weights (nset, cout, cin, 5, 5) -> [set][group order index i4][quad][lane][4]: slot 4 q + e = 3 j + mt
__global__ void pack_weight_mfma4_kernel(const float *__restrict__ w, float *__restrict__ packed, int cin, int cout,
                                         int ngroup, int slack, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int quads = quads4_of(cin), red = cin * KK;
  const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
  long long r = i >> 8;
  const int quad = (int)(r % quads);
  r /= quads;
  const int i4 = (int)(r & 15), set = (int)(r >> 4);
  const int slot = quad * 4 + e;
  float v = 0.f;
  if (slot < slots4_of(cin)) {
    const int j = slot / kMT, mt = slot - j * kMT;
    const int l0 = 4 * bitrev2(i4 & 3) + bitrev2(i4 >> 2);
    const int kk = l0 + kDcls[lane >> 4] + kWave * j;
    const int out = 16 * mt + (lane & 15);
    if (kk < red && out < cout) {
      const int tc = out / GO, o = out - tc * GO, group_in = cin / ngroup;
      const int tap = kk / cin, ci = kk - tap * cin;
      const int kh = tap / K5, kw = tap - kh * K5;
      const bool ok = (2 * HALF - kh - kw) * group_in - ci + (tc + slack) * group_in > 0;
      if (ok) v = w[(((size_t)set * ngroup + tc) * GO + o) * red + ci * KK + tap];
    }
  }
  packed[i] = v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CIN, int WAVES>
__global__ __launch_bounds__(WAVES * kWave, 2) void ee_conv_bulk_mfma4_kernel(
    EeGeom g, const int4 *__restrict__ blocks, int rp_n, int ct_n, const float *__restrict__ x, int shared_input,
    const float *__restrict__ wfrag, const float *__restrict__ bias, const float *__restrict__ slope,
    const float *__restrict__ residual, float *__restrict__ y, int pad_out, int s_lo, int s_hi) {
  constexpr int ITER = iter_of(CIN), QUADS = quads4_of(CIN), FRAG = frag4_floats(CIN);
  constexpr int COUT = 3 * (CIN == 14 ? 14 : CIN / 3);
  static_assert(COUT <= 16 * kMT && ITER <= 20 && K5 * CIN > 16, "shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *patch = smem;
  typedef const __attribute__((address_space(4))) int32_t const_i32_t;
  const_i32_t *brec = (const_i32_t *)(blocks + blockIdx.x);
  const int tile = brec[0], row0 = brec[1], col0 = brec[2];
  const int pn = blockIdx.y;
  const int set = pn / g.nimg;
  const int h = g.h, w = g.w;
  const int BR = rp_n, BC = 16 * ct_n, PW = BC + 4, PR = BR + 4;
  const int width = ((const_i32_t *)g.widths)[tile];
  {
    const int cmax = (col0 + BC < width ? col0 + BC : width) - 1;
    const int pmin = tile * h + row0 + col0, pmax = tile * h + row0 + BR - 1 + cmax;
    if (pmax + g.ngroup - 1 < s_lo || pmin >= s_hi) return;
  }
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  {
    // the patch by LDS-DMA (see ee_conv_bulk_mfma_kernel)
    typedef __attribute__((address_space(3))) void lds_ptr_t;
    typedef const __attribute__((address_space(1))) void glb_ptr_t;
    const int xi = shared_input ? pn % g.nimg : pn;
    const size_t tile_elems = (size_t)(h + 2 * PAD) * (w + 2 * PAD) * CIN;
    const float *xt = x + ((size_t)xi * g.npart + tile) * tile_elems + ((size_t)row0 * (w + 2 * PAD) + col0) * CIN;
    const int row16 = PW * CIN / 4;
    const int lim16 = (w + 2 * PAD - col0) * CIN / 4;
    const int npiece = PR * row16;
    const size_t buf_pitch = (size_t)(w + 2 * PAD) * CIN;
    int pr = tid / row16, j = tid - pr * row16;
    const int dpr = (WAVES * kWave) / row16, dj = (WAVES * kWave) - dpr * row16;
    for (int p0 = 0; p0 < npiece; p0 += WAVES * kWave) {
      const bool in = p0 + tid < npiece;
      const int prc = in ? pr : PR - 1, jc = in ? j : row16 - 1;
      const float *src = xt + prc * buf_pitch + 4 * (jc < lim16 ? jc : lim16 - 1);
      __builtin_amdgcn_global_load_lds((glb_ptr_t *)src, (lds_ptr_t *)(patch + (size_t)(p0 + wave * kWave) * 4), 16, 0, 0);
      pr += dpr;
      j += dj;
      if (j >= row16) {
        j -= row16;
        pr++;
      }
    }
    const int rounded = (npiece + WAVES * kWave - 1) / (WAVES * kWave) * (WAVES * kWave) * 4;
    for (int k = tid; k < kPatchSlack; k += WAVES * kWave) patch[rounded + k] = 0.f;
  }
  const int rp = wave % rp_n, ct = wave / rp_n;
  const int q = lane >> 4;  // block of the operands = lane class of the group; row quad of the results
  const bool live = col0 + 16 * ct < width;
  // the first group's weight pieces and the residual of this wave's outputs travel with the patch
  const float *wset = wfrag + (size_t)set * 16 * FRAG;
  float4 aq[QUADS];
  auto load_a = [&](int i4, int qd) { aq[qd] = *(reinterpret_cast<const float4 *>(wset + (size_t)i4 * FRAG) + qd * kWave + lane); };
#pragma unroll
  for (int qd = 0; qd < QUADS; qd++) load_a(0, qd);
  // way out: lane L holds outputs 16 mt + 4 (L >> 4) + r of position (row rp, column L & 15)
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int row = row0 + rp, col = col0 + 16 * ct + (lane & 15);
  const int colc = col < width ? col : width - 1;  // (lanes past the tile's width: a live address, nothing stored)
  const size_t ob0 = ((((size_t)pn * g.npart + tile) * (h + 2 * pad_out) + row + pad_out) * (w + 2 * pad_out) + colc + pad_out) * COUT;
  f2 r01[kMT], r23[kMT];
#pragma unroll
  for (int mt = 0; mt < kMT; mt++) {
    r01[mt] = r23[mt] = (f2){0.f, 0.f};
    if (residual) {
      const int out0 = 16 * mt + 4 * q;
      r01[mt] = *reinterpret_cast<const f2 *>(residual + ob0 + (out0 < COUT ? out0 : COUT - 2));
      r23[mt] = *reinterpret_cast<const f2 *>(residual + ob0 + (out0 + 2 < COUT ? out0 + 2 : COUT - 2));
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the patch DMA of every wave
  if (!live) return;                                                         // (no barrier from here on)
  const int dcls = q == 0 ? kDcls[0] : (q == 1 ? kDcls[1] : (q == 2 ? kDcls[2] : kDcls[3]));
  const int kh_stride = (PW - K5) * CIN;
  // byte offset of entry kk = l0 + dcls + 64 j of this lane's window: 4 (lane_base + kk + kh kh_stride), kh = the
  // window row of entry dcls + 64 j (compile-time per block: packed three bits per step), one row further from
  // l0 = thr on in the ONE step of this block whose sixteen classes straddle a row end (5 CIN = 210 entries)
  const unsigned base_b = 4u * (unsigned)(((rp)*PW + 16 * ct + (lane & 15)) * CIN + dcls);
  unsigned long long khpack = 0;
  int jspec = -1, thr = 99;
#pragma unroll
  for (int j = 0; j < ITER; j++) {
    const int c = dcls + kWave * j;
    int kh0 = c / (K5 * CIN);
    kh0 = kh0 < K5 - 1 ? kh0 : K5 - 1;
    khpack |= (unsigned long long)kh0 << (3 * j);
    const int cross = (kh0 + 1) * K5 * CIN - c;
    if (kh0 < K5 - 1 && cross < 16) jspec = j, thr = cross;
  }
  float bj[ITER];
  auto load_b = [&](int l0, int j) {
    int kh = (int)((khpack >> (3 * j)) & 7u);
    if (j == jspec && l0 >= thr) kh++;
    const unsigned off = base_b + 4u * (unsigned)(kWave * j + l0) + 4u * (unsigned)(kh * kh_stride);
    bj[j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(patch) + off);
  };
#pragma unroll
  for (int j = 0; j < ITER; j++) load_b(0, j);

  f32x4 T1[kMT], T2[kMT], T3[kMT], T4[kMT], tot[kMT];
#pragma unroll
  for (int t = 0; t < kMT; t++) T1[t] = T2[t] = T3[t] = T4[t] = tot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int a = 0; a < 4; a++) {
    f32x4 Q[kMT];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int i4 = 4 * a + b;
      // the next group (behind the last one: a dummy, group 15 again)
      const int n4 = i4 + 1 < 16 ? i4 + 1 : 15;
      const int ln = b < 3 ? 4 * bitrev2(b + 1) + bitrev2(a) : (a < 3 ? bitrev2(a + 1) : 4 * bitrev2(3) + bitrev2(3));
      f32x16 acc[kMT];
#pragma unroll
      for (int t = 0; t < kMT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
      for (int j = 0; j < ITER; j++) {
#pragma unroll
        for (int mt = 0; mt < kMT; mt++) {
          const int s = j * kMT + mt;
          const float av = s % 4 == 0 ? aq[s / 4].x : (s % 4 == 1 ? aq[s / 4].y : (s % 4 == 2 ? aq[s / 4].z : aq[s / 4].w));
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x1f32(av, bj[j], acc[mt], 0, 0, 0);
          if (s % 4 == 3 || s == ITER * kMT - 1) {  // (this piece's last step has issued)
            load_a(n4, s / 4);
            __builtin_amdgcn_sched_barrier(0);  // here, not where the scheduler would cluster the fetches: a piece is
          }                                     // waited for by count, in the order of its use
        }
        load_b(ln, j);
      }
      // the butterfly: xor 32 and xor 16 inside the lane, then the stack over the group index
#pragma unroll
      for (int t = 0; t < kMT; t++) {
        f32x4 s2;
#pragma unroll
        for (int r = 0; r < 4; r++) s2[r] = (acc[t][r] + acc[t][4 + r]) + (acc[t][8 + r] + acc[t][12 + r]);
        if ((b & 1) == 0) {
          T1[t] = s2;
        } else {
          const f32x4 u = T1[t] + s2;  // xor 8
          if ((b & 2) == 0)
            T2[t] = u;
          else
            Q[t] = T2[t] + u;  // xor 4
        }
      }
    }
    // xor 2, xor 1 over the outer index: selects, not branches (one basic block per turn of the loop -- behind
    // branches the wait-count pass drains every fetch at the loop's head); only the last turn's `tot` is used
    const bool even = (a & 1) == 0, second = a == 1;
#pragma unroll
    for (int t = 0; t < kMT; t++) {
      const f32x4 u = T3[t] + Q[t];
      tot[t] = T4[t] + u;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        T3[t][r] = even ? Q[t][r] : T3[t][r];
        T4[t][r] = second ? u[r] : T4[t][r];
      }
    }
  }
  // way out (bias, slope, the residual fetched in the prologue), 8-byte stores
  if (col >= width) return;
  const bool whole = s_lo <= 0 && s_hi >= g.h * g.npart + g.w + g.ngroup - 2;
  const float *bset = bias + set * COUT, *sset = slope ? slope + set * COUT : nullptr;
  const int plane = tile * h + row + col;
#pragma unroll
  for (int mt = 0; mt < kMT; mt++) {
    const int out0 = 16 * mt + 4 * q;
    if (out0 >= COUT) continue;
    const bool second = out0 + 2 < COUT;
    const f2 b01 = *reinterpret_cast<const f2 *>(bset + out0);
    const f2 b23 = second ? *reinterpret_cast<const f2 *>(bset + out0 + 2) : (f2){0.f, 0.f};
    f2 s01 = {1.f, 1.f}, s23 = {1.f, 1.f};
    if (sset) {
      s01 = *reinterpret_cast<const f2 *>(sset + out0);
      if (second) s23 = *reinterpret_cast<const f2 *>(sset + out0 + 2);
    }
    const size_t ob = ob0 + out0;
    const f32x4 t = tot[mt];
    float v[4] = {t[0] + b01[0], t[1] + b01[1], t[2] + b23[0], t[3] + b23[1]};
    const float sl[4] = {s01[0], s01[1], s23[0], s23[1]}, rs[4] = {r01[mt][0], r01[mt][1], r23[mt][0], r23[mt][1]};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (v[r] < 0) v[r] = v[r] * sl[r];
      if (residual) v[r] = v[r] + rs[r];
    }
    if (whole) {
      *reinterpret_cast<f2 *>(y + ob) = (f2){v[0], v[1]};
      if (second) *reinterpret_cast<f2 *>(y + ob + 2) = (f2){v[2], v[3]};
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int step = plane + (out0 + r) / GO;
        if (out0 + r < COUT && step >= s_lo && step < s_hi) y[ob + r] = v[r];
      }
    }
  }
}

}  // namespace

int ee_mfma_packed_floats(int nset, int cin) { return nset * 64 * frag_floats(cin); }

int ee_pack_weight_mfma(const float *w, float *packed, int nset, int cout, int cin, int ngroup, int constrain,
                        void *stream) {
  PCONV_REQUIRE(cout == GO * ngroup && cin % ngroup == 0 && (constrain == 5 || constrain == 6) && cout <= 16 * kMT,
                "ee_pack_weight_mfma: bad layer shape");
  const long long total = (long long)ee_mfma_packed_floats(nset, cin);
  hipLaunchKernelGGL(pack_weight_mfma_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w,
                     packed, cin, cout, ngroup, constrain == 5 ? 0 : 1, total);
  PCONV_LAUNCH_CHECK("ee_pack_weight_mfma");
  return PCONV_OK;
}

// a block is nt * rp_n rows x 16 ct_n columns, rp_n * ct_n = waves of a workgroup, nt = rows of a wave (1 or 2;
// PCONV_EE_MFMA_NT); rows per tile must be a multiple of nt * rp_n
int ee_mfma4_packed_floats(int nset, int cin) { return nset * 16 * frag4_floats(cin); }

int ee_pack_weight_mfma4(const float *w, float *packed, int nset, int cout, int cin, int ngroup, int constrain,
                         void *stream) {
  PCONV_REQUIRE(cout == GO * ngroup && cin % ngroup == 0 && (constrain == 5 || constrain == 6) && cout <= 16 * kMT,
                "ee_pack_weight_mfma4: bad layer shape");
  const long long total = (long long)ee_mfma4_packed_floats(nset, cin);
  hipLaunchKernelGGL(pack_weight_mfma4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w,
                     packed, cin, cout, ngroup, constrain == 5 ? 0 : 1, total);
  PCONV_LAUNCH_CHECK("ee_pack_weight_mfma4");
  return PCONV_OK;
}

// the four-classes-per-instruction form (42 channels, one row per wave)
int ee_conv_bulk_mfma4(const EeGeom *g, const void *blocks, int nblocks, int rp_n, int ct_n, int waves, const float *x,
                       const float *wfrag4, const float *bias, const float *slope, const float *residual, float *y,
                       int cin, int cout, int pad_out, int s_lo, int s_hi, void *stream) {
  PCONV_REQUIRE(cin == 42 && cout == 42 && g->ngroup == 14, "ee_conv_bulk_mfma4: 42 -> 42 channels only");
  PCONV_REQUIRE(rp_n > 0 && ct_n > 0 && rp_n * ct_n == waves && (waves == 4 || waves == 8) && g->h % rp_n == 0,
                "ee_conv_bulk_mfma4: bad block shape");
  PCONV_REQUIRE(s_lo < s_hi && nblocks > 0, "ee_conv_bulk_mfma4: bad range");
  const size_t round = (size_t)waves * kWave * 16;
  const size_t patch_bytes = ((size_t)(rp_n + 4) * (16 * ct_n + 4) * cin * sizeof(float) + round - 1) / round * round;
  const size_t smem = patch_bytes + kPatchSlack * sizeof(float);
  PCONV_REQUIRE(smem <= 160 * 1024, "ee_conv_bulk_mfma4: block needs %zu bytes of LDS", smem);
  const dim3 grid((unsigned)nblocks, (unsigned)(3 * g->nimg));
  PCONV_REQUIRE(grid.y <= 65535u, "ee_conv_bulk_mfma4: too many images for one launch");
  typedef void (*kernel_t)(EeGeom, const int4 *, int, int, const float *, int, const float *, const float *, const float *,
                           const float *, float *, int, int, int);
  static const kernel_t kernels[2] = {ee_conv_bulk_mfma4_kernel<42, 4>, ee_conv_bulk_mfma4_kernel<42, 8>};
  const int kind = waves == 8;
  {
    static std::atomic<unsigned long long> raised[2];
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    const unsigned long long bit = 1ULL << (device & 63);
    if (!(raised[kind].load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernels[kind]),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        pconv_set_error("ee_conv_bulk_mfma4: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        return PCONV_ELAUNCH;
      }
      raised[kind].fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kernels[kind], grid, dim3(waves * kWave), smem, as_stream(stream), *g, (const int4 *)blocks, rp_n, ct_n, x,
                     0, wfrag4, bias, slope, residual, y, pad_out, s_lo, s_hi);
  PCONV_LAUNCH_CHECK("ee_conv_bulk_mfma4");
  return PCONV_OK;
}

// PCONV_EE_MFMA_WSRC=ring: the LDS-ring form of the one-row kernel (default: direct fetch)
static bool mfma_direct(int nt) {
  const bool ring = getenv("PCONV_EE_MFMA_WSRC") && getenv("PCONV_EE_MFMA_WSRC")[0] == 'r';  // (per call: tests switch it)
  return nt == 1 && !ring;
}

int ee_mfma_block_shape(int h, int cin, int *rp_n, int *ct_n, int *waves, int *nt) {
  const int wv = getenv("PCONV_EE_MFMA_WAVES") ? atoi(getenv("PCONV_EE_MFMA_WAVES")) : 4;  // (per engine)
  // measured (MI355X, 4096x2048, one frame x 3 sets per launch, profiles/round5_entropy_mfma_variants.txt): one row per
  // wave 452 us per full launch (two rows: 474; eight waves per workgroup: 509 / 550), 591 / 715 us for a frame
  // in four step ranges
  const int nt_env = getenv("PCONV_EE_MFMA_NT") ? atoi(getenv("PCONV_EE_MFMA_NT")) : 1;
  const int nw = wv == 8 ? 8 : 4;
  const int n = (nt_env == 1 || (h & 1)) ? 1 : 2;
  if ((cin != 42 && cin != 14) || h < n) return 0;
  const int rows = h / n;  // wave rows per tile
  int rp = rows;
  const int cap = nw == 8 ? 4 : (n == 1 ? 4 : 2);
  while (rp > cap || rows % rp || nw % rp) rp--;
  *rp_n = rp;
  *ct_n = nw / rp;
  *waves = nw;
  *nt = n;
  return 1;
}

int ee_conv_bulk_mfma(const EeGeom *g, const void *blocks, int nblocks, int rp_n, int ct_n, int waves, int nt, const float *x,
                      int shared_input, const float *wfrag, const float *bias, const float *slope,
                      const float *residual, float *y, int cin, int cout, int pad_out, int s_lo, int s_hi,
                      void *stream) {
  PCONV_REQUIRE((cin == 42 || cin == 14) && cout == 42 && g->ngroup == 14, "ee_conv_bulk_mfma: 14 / 42 -> 42 channels only");
  PCONV_REQUIRE(cin == 42 || (nt == 1 && mfma_direct(nt)), "ee_conv_bulk_mfma: the input layer takes the one-row direct form");
  PCONV_REQUIRE(rp_n > 0 && ct_n > 0 && rp_n * ct_n == waves && (waves == 4 || waves == 8) && (nt == 1 || nt == 2) &&
                    g->h % (nt * rp_n) == 0,
                "ee_conv_bulk_mfma: bad block shape");
  PCONV_REQUIRE(s_lo < s_hi && nblocks > 0, "ee_conv_bulk_mfma: bad range");
  // ring + patch, the patch rounded up to whole DMA rounds of the workgroup (16 bytes per thread)
  const size_t round = (size_t)waves * kWave * 16;
  const size_t patch_bytes = ((size_t)(nt * rp_n + 4) * (16 * ct_n + 4) * cin * sizeof(float) + round - 1) / round * round;
  const bool direct = mfma_direct(nt);
  const size_t smem = (direct ? 0 : (size_t)kRing * frag_floats(42) * sizeof(float)) + patch_bytes + kPatchSlack * sizeof(float);
  PCONV_REQUIRE(smem <= 160 * 1024, "ee_conv_bulk_mfma: block needs %zu bytes of LDS", smem);
  const dim3 grid((unsigned)nblocks, (unsigned)(3 * g->nimg));
  PCONV_REQUIRE(grid.y <= 65535u, "ee_conv_bulk_mfma: too many images for one launch");
  typedef void (*kernel_t)(EeGeom, const int4 *, int, int, const float *, int, const float *, const float *, const float *,
                           const float *, float *, int, int, int);
  const int kind = cin == 14 ? 6 + (waves == 8) : direct ? 4 + (waves == 8) : (waves == 8 ? 2 : 0) + (nt == 1 ? 1 : 0);
  static const kernel_t kernels[8] = {ee_conv_bulk_mfma_kernel<42, 4, 2, false>, ee_conv_bulk_mfma_kernel<42, 4, 1, false>,
                                      ee_conv_bulk_mfma_kernel<42, 8, 2, false>, ee_conv_bulk_mfma_kernel<42, 8, 1, false>,
                                      ee_conv_bulk_mfma_kernel<42, 4, 1, true>,  ee_conv_bulk_mfma_kernel<42, 8, 1, true>,
                                      ee_conv_bulk_mfma_kernel<14, 4, 1, true>,  ee_conv_bulk_mfma_kernel<14, 8, 1, true>};
  {
    // the dynamic-LDS limit is a per-device attribute of the function (conv.hip)
    static std::atomic<unsigned long long> raised[8];
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    const unsigned long long bit = 1ULL << (device & 63);
    if (!(raised[kind].load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernels[kind]),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        pconv_set_error("ee_conv_bulk_mfma: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        return PCONV_ELAUNCH;
      }
      raised[kind].fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kernels[kind], grid, dim3(waves * kWave), smem, as_stream(stream), *g, (const int4 *)blocks, rp_n, ct_n, x,
                     shared_input, wfrag, bias, slope, residual, y, pad_out, s_lo, s_hi);
  PCONV_LAUNCH_CHECK("ee_conv_bulk_mfma");
  return PCONV_OK;
}
