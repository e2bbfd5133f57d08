// ARCHIVED EXPERIMENT (round 4) -- not built, not part of the product.
//
// "Streamed" 1x1 convolution: one persistent workgroup per CU, eight matrix waves that never leave the reduction loop
// (three-stage LDS-DMA ring, counted waits) park each tile's accumulators in LDS, eight drain waves take the parked
// tile out in 16-byte quads under the next tile's matrix loop.  Built end to end in csrc/conv.hip (this is the code as
// it stood there, with the dispatch branches at the end), bit-identical to the tiled kernel on every shape of
// tests/test_gpu_ops.py::test_streamed_1x1_equals_tiled_kernel (commit "Quad way out for the stride-2 3x3 layers
// too" still contains it), measured with in-kernel stamps (profiles/round4_stream_1x1_*.txt) and withdrawn: level
// with the tiled kernel (96->192 + residual 2.04 vs 2.06 ms, 192->96 1.68 vs 1.72, 192->192 + residual 2.96 vs
// 2.72, GDN 1.84-1.90 vs 1.84-1.89 at the codec's batch) -- a chunk period takes 5 000-5 700 cycles for 3 072 of
// matrix work (block of two waves 3 900, chunk barrier + loop 1 400-1 800), the same ~0.55 the tiled kernel's two
// workgroups per CU reach.  What it taught, and what went into the product instead, is the quad way out
// (conv_epilogue_quads in csrc/conv.hip): the element-wise way out was instruction-bound, not latency-bound.
// DESIGN.md section 4 has the write-up (including three ways the compiler's s_waitcnt insertion defeats a register
// ring of loads in flight).
//
// It needs conv.hip's ConvCfg / ConvStager / conv_chunk_step / static_for_ (then named stream_static_for) and the
// load4_base_off / store4_base_off helpers around it.

// the matrix block of the streamed kernel: conv_chunk_step with the LDS-DMA pieces of the stage requested in this
// period issued in the shadow of the k-pairs' MFMAs (one workgroup per CU: both waves of a SIMD leave the chunk
// barrier together, and ~100 address / branch instructions in front of the first MFMA left the matrix pipe idle)
template <class C, int J, int J1>
__device__ __forceinline__ void stream_issue(const ConvStager<C> &st) {
  if constexpr (J < J1) {
    st.template issue<J>();
    stream_issue<C, J + 1, J1>(st);
  }
}

template <class C, bool SQ, int KP>
__device__ __forceinline__ void stream_chunk_step(f32x16 (&acc)[C::MTv][C::NTv], float (&a)[kAhead + 1][C::MTv],
                                                  float (&b)[kAhead + 1][C::NTv], unsigned abase,
                                                  const unsigned (&bbase)[C::NTv][C::ND], const ConvStager<C> &st) {
  constexpr int NP = C::KK / 2, SETS = kAhead + 1, PIECES = ConvStager<C>::PIECES;
  if constexpr (KP + kAhead < NP)
    conv_read_pair<C, KP + kAhead>(a[(KP + kAhead) % SETS], b[(KP + kAhead) % SETS], abase, bbase);
  constexpr int ahead = (NP - 1 - KP) < kAhead ? (NP - 1 - KP) : kAhead;
  float(&A)[C::MTv] = a[KP % SETS];
  float(&B)[C::NTv] = b[KP % SETS];
  conv_wait_pair<ahead * (C::MTv + C::NTv)>(A, B);
  if (SQ) {
#pragma unroll
    for (int n = 0; n < C::NTv; n++) B[n] = B[n] * B[n];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < C::MTv; m++)
#pragma unroll
    for (int n = 0; n < C::NTv; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m], B[n], acc[m][n], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  stream_issue<C, KP * PIECES / NP, (KP + 1) * PIECES / NP>(st);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KP + 1 < NP) stream_chunk_step<C, SQ, KP + 1>(acc, a, b, abase, bbase, st);
}

// (measured: 192->96 0.31 instead of 0.25 ms, the layers with a residual 3-5 % slower: off)
#ifndef PCONV_STREAM_DMA_INTERLEAVE
#define PCONV_STREAM_DMA_INTERLEAVE 0
#endif
#ifdef PCONV_STREAM_STAMP
// profiling build: cycle counter of one workgroup's waves at the phases of its first 64 chunk periods:
// matrix waves [period start, matrix block issued, stage wait over, barrier passed], drain waves [period start,
// slice written, barrier passed, -]
__device__ unsigned long long stream_stamps[16][64][4];
#define STREAM_STAMP_AT(w, k)                                                                          \
  if (blockIdx.x == gridDim.x / 2 && stamp_period < 64 && (threadIdx.x & 63) == 0)                      \
  stream_stamps[w][stamp_period][k] = __builtin_readcyclecounter()
#else
#define STREAM_STAMP_AT(w, k)
#endif
#define STREAM_STAMP(k) STREAM_STAMP_AT(wave, k)
#ifndef PCONV_STREAM_DEPTH
#define PCONV_STREAM_DEPTH 1
#endif
#ifndef PCONV_STREAM_DRAIN_WAVES
#define PCONV_STREAM_DRAIN_WAVES 8
#endif
// ---- streamed 1x1 convolution: matrix waves + drain waves ----------------------------------
// The 1x1 / GDN layers move 1.1-3 KB per pixel through HBM for 37-74 KFLOP: about as far from the HBM roof as from
// the matrix roof, and in the tiled kernel a workgroup is a matrix phase (~100 TFLOP/s while it lasts) FOLLOWED by a
// memory phase (~5.4 TB/s while it lasts); two workgroups per CU do not interleave them (profiles/round3_1x1_*.txt,
// bench.py's two-roof class table: mfma 0.42 + hbm 0.43 for 96->192 + residual).  Here the phases run side by side
// inside ONE persistent workgroup of twelve waves:
//   * waves 0-7 (matrix waves) are the tiled kernel's reduction loop and nothing else: LDS-DMA stages, counted LDS
//     operand reads, 3 MFMAs per k-pair -- tile after tile without leaving the loop.  After the last chunk of a tile
//     they park its accumulators in an LDS tile [BM couts][PX pixels] (48 ds_write per lane) and go on;
//   * waves 8-11 (drain waves) take the parked tile out while the matrix waves multiply the next one: per lane 24
//     quads of 4 consecutive pixels -- one 16-byte LDS read, 16-byte loads of the residual / the GDN's own input
//     (requested one slice ahead and held in registers across the chunk barriers: the drain waves' barrier is a bare
//     s_barrier, the compiler's vmcnt(n) in front of the uses are the only waits), the epilogue arithmetic of
//     conv_epilogue_pipe in the same order, one 16-byte store;
//   * the chunk barriers are the only synchronisation: the drain of tile k is spread over the first NCH - 1 chunks
//     of tile k + 1, the last chunk's period is the matrix waves' parking slot.
// Same MFMA chain and the same epilogue operations per output as the tiled kernel: identical bits.
template <int WM, int WN, int NCH, bool SQ, bool RES>
__global__ __launch_bounds__(512 + 64 * PCONV_STREAM_DRAIN_WAVES, (8 + PCONV_STREAM_DRAIN_WAVES) / 4) void conv1x1_stream_kernel(
    const float *__restrict__ in, const float *__restrict__ wp, float *__restrict__ out, int h, int w, int cout,
    int cout_pad, int tiles_r, int tiles_c, int cblocks, int ntiles, ConvView vin, ConvView vout, ConvEpilogue ep) {
  using C = ConvCfg<3, 1, WM, WN, 1, 1, PCONV_KC1>;
  using P = typename C::P;
  constexpr int MT = 3, NT = 1, KC = PCONV_KC1;
  constexpr int BM = C::BM, PX = 32 * WN, ROWS = C::ROWS;
  constexpr int kDrain = 64 * PCONV_STREAM_DRAIN_WAVES;  // drain lanes
  // LDS-DMA stages of the reduction loop: a ring of three where it fits (a stage is requested two chunks ahead:
  // the input tile of a 1x1 layer comes from HBM and one workgroup per CU has nobody to hide a late stage behind)
  constexpr int RING = (3 * C::STAGE + BM * PX + 2 * BM) * 4 <= 160 * 1024 ? 3 : 2;
  constexpr int NQ = BM * PX / 4 / kDrain;             // quads per drain lane and tile
  constexpr int QP = (NQ + NCH - 2) / (NCH - 1);       // ... and per chunk period (the last period drains nothing)
  static_assert(BM * PX / 4 % kDrain == 0 && PX % 64 == 0, "drain quads divide evenly");
  extern __shared__ float lds[];
  float *ot = lds + RING * C::STAGE;   // parked accumulators [BM][PX]
  float *bias_s = ot + BM * PX;        // [BM] (one cout block: cout == BM)
  float *slope_s = bias_s + BM;        // [BM]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int32_t *__restrict__ col_limit = ep.col_limit;
  const int npart = ep.npart;

  struct Tile {
    int cb, t, r0, c0;
  };
  auto tile_of = [&](int b) {
    Tile q;
    q.cb = b % cblocks;
    b /= cblocks;
    q.r0 = (b % tiles_r) * ROWS;
    b /= tiles_r;
    q.c0 = (b % tiles_c) * kTileCols;
    q.t = b / tiles_c;
    return q;
  };
  auto dead = [&](const Tile &q) { return col_limit && q.c0 >= col_limit[q.t % npart]; };
  // first live tile of this workgroup's list (b, b + G, ...) at or after b
  auto next_live = [&](int b) {
    while (b < ntiles && dead(tile_of(b))) b += G;
    return b;
  };

  if (wave < 8) {
    // ---------------- matrix waves ----------------
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    unsigned xoffs[C::XLD], woffs[C::WLD];
#pragma unroll
    for (int j = 0; j < C::WLD; j++) {
      int e4 = tid + j * C::THREADS;
      e4 = e4 < C::WSZ / 4 ? e4 : 0;
      const int kk = e4 / (BM / 4);
      const int co = (e4 % (BM / 4)) * 4;
      woffs[j] = (unsigned)(kk * cout_pad + co) * 4u;
    }
    // what the NEXT stage request reads: uniform bases (advanced by one chunk per request) + the lanes' offsets
    const float *xb = in, *wb = wp;
    auto set_tile = [&](const Tile &q) {
#pragma unroll
      for (int j = 0; j < C::XLD; j++) {
        const int e = tid + j * C::THREADS;  // (< XSZ: whole waves, static_assert below)
        const int pc = e % P::PC;
        const int pr = (e / P::PC) % P::PR;
        const int ci = e / (P::PC * P::PR);
        int ir = q.r0 + pr, ic = q.c0 + pc;
        ir = ir < h ? ir : h - 1;
        ic = ic < w ? ic : w - 1;
        xoffs[j] = (unsigned)((ci * vin.cs + (long long)ir * vin.rs + ic) * 4);
      }
      xb = in + (size_t)q.t * vin.ts;
      wb = wp + q.cb * BM;
    };
    const size_t xstep = (size_t)KC * vin.cs, wstep = (size_t)C::KK * cout_pad;
    // DMA pieces this wave issues per stage (the ragged last weight piece belongs to the first waves only): the
    // wait in front of a chunk barrier lets exactly the stages requested after the one about to be read stay in flight
    constexpr int WFULL = (C::WSZ / 4) / C::THREADS;                  // weight pieces every wave issues
    constexpr int WPART = (C::WSZ / 4) % C::THREADS / 64;             // waves that issue one more
    static_assert(C::XSZ % C::THREADS == 0 && (C::WSZ / 4) % 64 == 0, "stage pieces are whole waves");
    const bool extra = wave < WPART;
    // one stage = XLD patch dwords + WFULL (+ 1) weight float4s per lane, straight-line: no bounds, no ragged
    // channels (cin is a whole number of chunks), "scalar base + lane offset" addresses
    auto request_stage = [&](int slot) {
      float *xs = lds + slot * C::STAGE;
#pragma unroll
      for (int j = 0; j < C::XLD; j++) {
        unsigned off = xoffs[j];
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_global_load_lds((glb_ptr_t *)((glb_bytes_t *)xb + off), (lds_ptr_t *)(xs + j * C::THREADS + wave * 64),
                                         4, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < WFULL + 1; j++) {
        if (j < WFULL || extra) {
          unsigned off = woffs[j];
          asm volatile("" : "+v"(off));
          __builtin_amdgcn_global_load_lds((glb_ptr_t *)((glb_bytes_t *)wb + off),
                                           (lds_ptr_t *)(xs + C::XSZ + (j * C::THREADS + wave * 64) * 4), 16, 0, 0);
        }
      }
      xb += xstep;
      wb += wstep;
    };
    static_assert(C::WLD == WFULL + (WPART ? 1 : 0), "weight pieces per lane");
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);
    unsigned abase = lds0 + (unsigned)(C::XSZ + half * BM + wm * MT * 32 + l31) * 4u;
    unsigned bbase[NT][C::ND];
    {
      const int seg = wn;
      const int prow = seg >> 1, pcol = (seg & 1) * 32 + l31;
      bbase[0][0] = lds0 + (unsigned)(prow * P::PC + pcol + half * C::delta(0)) * 4u;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][0][r] = 0.f;

    int stamp_period = 0;
    (void)stamp_period;
    auto chunk_barrier = [&](int stages_behind) {
      // stages_behind: stages requested after the one the next chunk reads (0 .. RING - 2)
      if (stages_behind == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (extra) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::XLD + WFULL + 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::XLD + WFULL) : "memory");
      }
      STREAM_STAMP(2);
      asm volatile("s_barrier" ::: "memory");
      STREAM_STAMP(3);
      stamp_period++;
    };
    static_assert(RING == 2 || RING == 3, "ring of two or three stages");
    // Two cursors over this workgroup's live tiles: the REQUEST cursor (tile rb, chunk rc) runs RING - 1 chunks
    // ahead of the chunk being multiplied (tile b); stage s of the ring holds the chunks g with g % RING == s,
    // chunks counted across tiles.
    int b = next_live(blockIdx.x);
    int rb = b, rc = 0, rslot = 0;
    auto request_next = [&]() {
      if (rb >= ntiles) return false;
      request_stage(rslot);
      rslot = rslot == RING - 1 ? 0 : rslot + 1;
      if (++rc == NCH) {
        // (every stage of that tile has been requested: its offsets can go)
        rc = 0;
        rb = next_live(rb + G);
        if (rb < ntiles) set_tile(tile_of(rb));
      }
      return true;
    };
    if (b < ntiles) set_tile(tile_of(b));
    int behind = 0;
#pragma unroll
    for (int i = 0; i < RING - 1; i++) behind += request_next() ? 1 : 0;
    chunk_barrier(behind > 1 ? 1 : 0);  // B0: stage 0 has landed (a second requested stage may stay in flight)
    stamp_period = 0;
    int stage = 0;  // ring slot of the chunk being multiplied
    while (b < ntiles) {
#pragma unroll 1
      for (int chunk = 0; chunk < NCH; chunk++) {
        STREAM_STAMP(0);
        const bool issued = request_next();
        float a[kAhead + 1][MT], bq[kAhead + 1][NT];
        conv_chunk_prologue<C, 0>(a, bq, abase, bbase);
        conv_chunk_step<C, SQ, 0>(acc, a, bq, abase, bbase);
        STREAM_STAMP(1);
        const unsigned hop = stage == RING - 1 ? (unsigned)(-(RING - 1) * C::STAGE * 4) : (unsigned)(C::STAGE * 4);
        abase += hop;
        bbase[0][0] += hop;
        stage = stage == RING - 1 ? 0 : stage + 1;
        if (chunk == NCH - 1) {
          // park the tile: reg r of a 32x32 tile = cout row (r&3) + 8*(r>>2) + 4*half, pixel l31 of segment wn
#pragma unroll
          for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
              const int col = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              ot[col * PX + wn * 32 + l31] = acc[m][0][r];
              acc[m][0][r] = 0.f;
            }
        }
        // the next chunk's stage was requested RING - 1 chunks ago; what was requested since may stay in flight
        chunk_barrier(RING == 3 && issued ? 1 : 0);
      }
      b = next_live(b + G);
    }
    return;  // (a wave that has ended no longer counts at s_barrier: the drain waves finish alone)
  }

  // ---------------- drain waves ----------------
  const int dl = tid - 512;  // 0..kDrain-1
  for (int i = dl; i < BM; i += kDrain) {  // (cout == BM: one cout block)
    bias_s[i] = ep.bias ? ep.bias[i] : 0.f;
    slope_s[i] = ep.act == 1 ? ep.slope[i] : 1.f;
  }
  __syncthreads();  // B0
  const int act = ep.act;
  const int wo = w, ho = h;
  // quad j of a lane: parked cout row CSTEP * j + col0, pixels px0 .. px0 + 3 (the same for all its quads)
  constexpr int QROW = PX / 4;           // quads per cout row
  constexpr int CSTEP = kDrain / QROW;   // cout rows the drain lanes cover per quad index
  static_assert(kDrain % QROW == 0, "a quad index covers whole cout rows");
  const int col0 = dl / QROW, px0 = (dl % QROW) * 4;
  const int prow = px0 / kTileCols, pcol = px0 % kTileCols;
  struct QuadIn {
    float4 r, x;
  };
  // a parked tile as the drain lanes see it: wave-uniform element offsets of the tile's corner (scalar
  // registers); a lane adds its own, tile-invariant, offsets of quad 0 (quad j: + j * CSTEP channel strides)
  struct Prev {
    Tile q;
    bool edge;
    int trim_at;
    long long b_out, b_res, b_in;
  };
  auto describe = [&](int b) {
    Prev p;
    p.q = tile_of(b);
    p.edge = p.q.c0 + kTileCols > wo || p.q.r0 + ROWS > ho;
    p.trim_at = ((ep.trim || act == 2 || act == 3) && col_limit) ? col_limit[p.q.t % npart] : wo;
    p.b_out = (long long)p.q.t * vout.ts + (long long)p.q.r0 * vout.rs + p.q.c0;
    p.b_res = RES ? (long long)p.q.t * ep.vres.ts + (long long)p.q.r0 * ep.vres.rs + p.q.c0 : 0;
    p.b_in = (long long)p.q.t * vin.ts + (long long)p.q.r0 * vin.rs + p.q.c0;
    return p;
  };
  // (element offsets for the edge path; BYTE offsets in 32 bits for the fast path: a quad's address is a uniform
  // base in scalar registers + the lane's one offset register -- no vector address arithmetic, nothing per quad
  // for the compiler to hoist out of the tile loop into registers it does not have)
  const long long l_out = (long long)col0 * vout.cs + (long long)prow * vout.rs + pcol;
  const long long l_res = RES ? (long long)col0 * ep.vres.cs + (long long)prow * ep.vres.rs + pcol : 0;
  const long long l_in = (long long)col0 * vin.cs + (long long)prow * vin.rs + pcol;
  const unsigned lb_out = (unsigned)(l_out * 4), lb_res = (unsigned)(l_res * 4), lb_in = (unsigned)(l_in * 4);
  const long long s_out = (long long)CSTEP * vout.cs, s_res = (long long)CSTEP * ep.vres.cs, s_in = (long long)CSTEP * vin.cs;
  // requests what quads [J0, J0 + QP) of tile p read from memory (an edge tile: from addresses clamped into the
  // tensors; its quads are read again and written element by element).  Every load issued here is consumed by
  // write() on EVERY static path (the edge path touches the registers too, the first tile of a workgroup is peeled
  // off the steady loop): a load the compiler can see pending on some path gets an s_waitcnt vmcnt(0) in front of
  // the next write to its register -- right behind the chunk barrier, a whole memory round trip per period
  // (measured: ~3 300 cycles of a 6 100-cycle period).
  auto request = [&](const Prev &p, auto j0c, QuadIn (&dst)[QP]) {
    constexpr int J0 = decltype(j0c)::value;
    // (an edge tile's quads are read again, element by element, when they are written: what is requested here is
    // only waited for, so every lane asks for the first four floats of the tile's image -- always inside the
    // tensor; the lane offsets are unsigned, a clamp to the left of the tile's corner cannot be expressed in them)
    const unsigned o_res = p.edge ? 0u : lb_res, o_in = p.edge ? 0u : lb_in;
    const long long t_res = p.edge ? (long long)p.q.t * ep.vres.ts : p.b_res;
    const long long t_in = p.edge ? (long long)p.q.t * vin.ts : p.b_in;
    const long long q_res = p.edge ? 0 : s_res, q_in = p.edge ? 0 : s_in;
#pragma unroll
    for (int e = 0; e < QP; e++) {
      if (J0 + e < NQ) {
        if (RES) dst[e].r = load4_base_off((global_bytes *)(ep.residual + t_res + (J0 + e) * q_res), o_res);
        if (SQ) dst[e].x = load4_base_off((global_bytes *)(in + t_in + (J0 + e) * q_in), o_in);
      }
    }
  };
  // (PRELU / TRIM as constants: the drain waves' VALU instructions compete with the matrix waves' MFMAs for the
  // SIMD's issue port -- ~25 cycles apiece while those stream -- so a tile without PReLU whose columns are all
  // alive, the common case, must not pay for either)
  auto finish = [&](float v, float xv, float rv, float sl, bool trimmed, auto preluc, auto trimc) {
    if (SQ) {
      const float nrm = sqrtf(v);
      v = act == 2 ? xv / nrm : xv * nrm;
    } else if (decltype(preluc)::value) {
      if (v < 0) v = v * sl;
    }
    if (RES) v = rv + v;
    if (decltype(trimc)::value && trimmed) v = 0.f;
    return v;
  };
  const float *ot_lane = ot + col0 * PX + px0;
  auto write_quads = [&](const Prev &p, auto j0c, const QuadIn (&src)[QP], auto preluc, auto trimc) {
    constexpr int J0 = decltype(j0c)::value;
    const int ocol = p.q.c0 + pcol;
#pragma unroll
    for (int e = 0; e < QP; e++) {
      if (J0 + e < NQ) {
        const int j = J0 + e;
        const int co = CSTEP * j + col0;
        const float4 o4 = *reinterpret_cast<const float4 *>(ot_lane + j * (CSTEP * PX));
        const float bco = bias_s[co];
        const float sl = decltype(preluc)::value ? slope_s[co] : 1.f;
        const float o[4] = {o4.x, o4.y, o4.z, o4.w};
        const float xs[4] = {src[e].x.x, src[e].x.y, src[e].x.z, src[e].x.w};
        const float rs[4] = {src[e].r.x, src[e].r.y, src[e].r.z, src[e].r.w};
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = finish(o[k] + bco, xs[k], rs[k], sl, ocol + k >= p.trim_at, preluc, trimc);
        store4_base_off(out + p.b_out + j * s_out, lb_out, make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  };
  auto write = [&](const Prev &p, auto j0c, const QuadIn (&src)[QP]) {
    constexpr int J0 = decltype(j0c)::value;
    const int orow = p.q.r0 + prow, ocol = p.q.c0 + pcol;
    if (!p.edge) {
      const bool trims = p.trim_at < p.q.c0 + kTileCols;  // (uniform: some column of the tile is dead)
      if (!SQ && act == 1) {
        if (trims)
          write_quads(p, j0c, src, std::true_type{}, std::true_type{});
        else
          write_quads(p, j0c, src, std::true_type{}, std::false_type{});
      } else {
        if (trims)
          write_quads(p, j0c, src, std::false_type{}, std::true_type{});
        else
          write_quads(p, j0c, src, std::false_type{}, std::false_type{});
      }
      return;
    }
    // a tile that hangs over the right / lower edge: element by element (what was requested ahead is only waited
    // for: see request())
#pragma unroll
    for (int e = 0; e < QP; e++) {
      if (J0 + e < NQ) {
        if (RES) asm volatile("" ::"v"(src[e].r.x), "v"(src[e].r.y), "v"(src[e].r.z), "v"(src[e].r.w));
        if (SQ) asm volatile("" ::"v"(src[e].x.x), "v"(src[e].x.y), "v"(src[e].x.z), "v"(src[e].x.w));
      }
    }
    if (orow >= ho) return;
#pragma unroll 1
    for (int e = 0; e < QP; e++) {
      const int j = J0 + e;
      if (j >= NQ) break;
      const int co = CSTEP * j + col0;
      const float *osrc = ot_lane + j * (CSTEP * PX);
      const float bco = bias_s[co], sl = slope_s[co];
      float *dst = out + p.b_out + l_out + j * s_out;
      const float *xsrc = in + p.b_in + l_in + j * s_in;
      const float *rsrc = RES ? ep.residual + p.b_res + l_res + j * s_res : nullptr;
#pragma unroll 1
      for (int k = 0; k < 4; k++) {
        if (ocol + k >= wo) break;
        const float xv = SQ ? xsrc[k] : 0.f;
        const float rv = RES ? rsrc[k] : 0.f;
        dst[k] = finish(osrc[k] + bco, xv, rv, sl, ocol + k >= p.trim_at, std::true_type{}, std::true_type{});
      }
    }
  };
  auto bare_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // What the quads read from memory is requested DEPTH chunk periods before they are written and waits in a
  // register ring.
  constexpr int DEPTH = PCONV_STREAM_DEPTH;
  static_assert(DEPTH >= 1 && DEPTH <= NCH - 1, "request distance in chunk periods");
  // The ring's slots rotate by NAME, not by copying: the period loops are unrolled (NCH is a multiple of the ring
  // size, so the rotation is the same in every tile) and slot (p + DEPTH) % SLOTS is requested in period p, slot
  // p % SLOTS written.  (Copying slot d + 1 to slot d at the end of a period made the compiler wait, right behind
  // the chunk barrier, for the loads it had just issued: every period paid a whole memory round trip, ~3 300 cycles.)
  constexpr int SLOTS = DEPTH + 1;
  static_assert(NCH % SLOTS == 0, "the ring rotation must close over a tile");
  QuadIn ring[SLOTS][QP];
  int stamp_period = 0;
  (void)stamp_period;
  // one period of the drain: P = chunk period inside the tile the matrix waves are on (`self`; SELF false: the
  // tail), `prev` = the parked tile (HAVE false: the workgroup's first tile, nothing parked yet)
  auto period_work = [&](auto pc, auto havec, auto selfc, const Prev &prev, const Prev &self) {
    constexpr int p = decltype(pc)::value, p2 = p + DEPTH;
    constexpr bool HAVE = decltype(havec)::value, SELF = decltype(selfc)::value;
    if constexpr (RES || SQ) {
      if constexpr (p2 < NCH - 1) {
        if constexpr (HAVE) request(prev, std::integral_constant<int, p2 * QP>{}, ring[p2 % SLOTS]);
      } else if constexpr (p2 >= NCH && SELF) {
        request(self, std::integral_constant<int, (p2 >= NCH ? p2 - NCH : 0) * QP>{}, ring[p2 % SLOTS]);
      }
    }
    if constexpr (HAVE && p < NCH - 1) write(prev, std::integral_constant<int, p * QP>{}, ring[p % SLOTS]);
  };
  auto tile_periods = [&](auto havec, auto selfc, const Prev &prev, const Prev &self) {
    constexpr bool SELF = decltype(selfc)::value;
    stream_static_for<0, SELF ? NCH : NCH - 1>([&](auto pc) {
      STREAM_STAMP(0);
      period_work(pc, havec, selfc, prev, self);
      STREAM_STAMP(1);
      if constexpr (SELF) {
        bare_barrier();
        STREAM_STAMP(2);
        stamp_period++;
      }
    });
  };
  auto zero_dead = [&](const Tile &q) {
    conv_zero_tile<BM, ROWS, kDrain>(out + (size_t)q.t * vout.ts, vout, 0, q.cb * BM, cout, q.r0, q.c0, ho, wo, dl);
  };
  // the workgroup's first live tile: nothing parked yet
  int b = blockIdx.x;
  for (; b < ntiles; b += G) {
    const Tile q = tile_of(b);
    if (!dead(q)) break;
    zero_dead(q);
  }
  if (b >= ntiles) return;
  Prev prev = describe(b);
  tile_periods(std::false_type{}, std::true_type{}, prev, prev);
  // steady state: tile `prev` is drained while the matrix waves are on tile `self`
  for (b += G; b < ntiles; b += G) {
    const Tile q = tile_of(b);
    if (dead(q)) {
      zero_dead(q);
      continue;
    }
    const Prev self = describe(b);
    tile_periods(std::true_type{}, std::true_type{}, prev, self);
    prev = self;
  }
  // the last tile, alone (no barriers: the matrix waves have ended)
  tile_periods(std::true_type{}, std::false_type{}, prev, prev);
}

template <int WM, int WN, int NCH, bool SQ, bool RES>
int launch_conv1x1_stream(const float *in, const float *wp, float *out, int tn, int h, int w, int cout, int cout_pad,
                          const ConvView &vin, const ConvView &vout, const ConvEpilogue &ep, hipStream_t stream) {
  using C = ConvCfg<3, 1, WM, WN, 1, 1, PCONV_KC1>;
  const int tiles_r = (h + C::ROWS - 1) / C::ROWS;
  const int tiles_c = (w + kTileCols - 1) / kTileCols;
  const int cblocks = cout / C::BM;
  const long long ntiles = (long long)tn * tiles_r * tiles_c * cblocks;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) {
    pconv_set_error("conv2d: %lld tiles out of range", ntiles);
    return PCONV_EINVAL;
  }
  constexpr int RING = (3 * C::STAGE + C::BM * 32 * WN + 2 * C::BM) * 4 <= 160 * 1024 ? 3 : 2;  // (as in the kernel)
  const size_t smem = ((size_t)RING * C::STAGE + (size_t)C::BM * 32 * WN + 2 * (size_t)C::BM) * sizeof(float);
  auto kern = conv1x1_stream_kernel<WM, WN, NCH, SQ, RES>;
  static std::atomic<unsigned long long> raised{0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) device = 0;
  const unsigned long long bit = 1ULL << (device & 63);
  static int cus[64] = {0};
  if (!(raised.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) {
      pconv_set_error("conv2d: cannot raise dynamic LDS to %zu: %s", smem, hipGetErrorString(e));
      return PCONV_ELAUNCH;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
    cus[device & 63] = n;
    raised.fetch_or(bit, std::memory_order_release);
  }
  // one persistent workgroup per CU (the parked tile + the stages fill its LDS), tiles dealt round-robin
  const int grid = (int)(ntiles < cus[device & 63] ? ntiles : cus[device & 63]);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512 + 64 * PCONV_STREAM_DRAIN_WAVES), smem, stream, in, wp, out, h, w, cout, cout_pad, tiles_r,
                     tiles_c, cblocks, (int)ntiles, vin, vout, ep);
  return PCONV_OK;
}

// the streamed form takes the 1x1 stride-1 layers of the codec's trunk: 96 or 192 input channels (6 / 12 chunks),
// 96 or 192 couts (one cout block), bias / PReLU / GDN / residual / trim on the way out
// (PCONV_CONV1X1=stream forces it where it applies, =tiled keeps the tiled kernel: A/B measurements, parity tests)
inline int stream_1x1_mode() {
  const char *env = getenv("PCONV_CONV1X1");  // (read per call: the parity test switches it)
  return env ? (env[0] == 's' ? 1 : (env[0] == 't' || env[0] == 'r' ? -1 : 0)) : 0;
}
inline bool use_stream_1x1(int cin, int cout, int tn, int h, int w) {
  const int mode = stream_1x1_mode();
  if (mode < 0) return false;
  if (!((cin == 96 || cin == 192) && PCONV_KC1 == 16 && (cout == 96 || cout == 192) && w >= 4)) return false;
  if (mode > 0) return true;
  return false;  // (until measured)
}

#ifdef PCONV_STREAM_STAMP
extern "C" int pconv_stream_read_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(stream_stamps), sizeof(stream_stamps)) == hipSuccess ? 0 : 1;
}
#endif


// ---- dispatch (pconv_conv2d / pconv_gdn) ----
#if 0
  } else if (k == 1 && stride == 1 && !gate && !d2w && act != 4 && use_stream_1x1(cin, cout, tn, h, w)) {
#define STREAM(WM, WN, NCH)                                                                                       \
  rc = residual ? launch_conv1x1_stream<WM, WN, NCH, false, true>(in, packed_w, out, tn, h, w, cout, cp, vin, vout, ep, s) \
                : launch_conv1x1_stream<WM, WN, NCH, false, false>(in, packed_w, out, tn, h, w, cout, cp, vin, vout, ep, s);
    if (cout == 96 && cin == 96) {
      STREAM(1, 8, 6)
    } else if (cout == 96) {
      STREAM(1, 8, 12)
    } else if (cin == 96) {
      STREAM(2, 4, 6)
    } else {
      STREAM(2, 4, 12)
    }
#undef STREAM
  if (use_stream_1x1(ch, ch, tn, h, w) && ch == 192)
    rc = residual ? launch_conv1x1_stream<2, 4, 12, true, true>(in, packed_gamma, out, tn, h, w, ch, cp, vin, vout, ep, s)
                  : launch_conv1x1_stream<2, 4, 12, true, false>(in, packed_gamma, out, tn, h, w, ch, cp, vin, vout, ep, s);
  else if (use_resident_1x1(ch, ch, tn, h, w) && ch > 96 && residual)
#endif
