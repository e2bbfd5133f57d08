// (r3: two kernels.  `ac_decode_kernel` is round 2's: it loads every CDF row and every stream byte from
// GLOBAL memory inside the serial loop, so its 377 ns/symbol is one memory round trip per symbol, not the
// decoder.  `ac_decode_lds_kernel` is the decoder itself: the rows of a step are known before its first
// symbol is decoded, so the whole step's rows and its byte window are copied to LDS by all lanes first
// (coalesced), the row of symbol i+1 is read while symbol i is decoded, and nothing in the dependent
// chain touches global memory.)
//
// How fast can ONE wave of the GPU run the codec's arithmetic decoder?  ("decoder on the
// device".)  The same decoder as csrc/coder.cpp -- 32-bit state, 8-symbol rows with total 65536, clz
// renormalisation, the symbol found as the number of thresholds (row[k] * range >> 16) <= offset -- written
// for a single wave: lane k evaluates threshold k (8 lanes work, the rest idle: the stream is serial), the
// state lives in scalar registers.  A stream encoded by libpconv_coder.so is decoded on the device, checked
// symbol by symbol, and timed with HIP events against the CPU decoder on the same stream.
//
//   hipcc --offload-arch=gfx950 -O3 -Iinclude tools/ac_device_probe.hip -o tools/_build/ac_device_probe \
//         -Lpseudocylindrical_convolution_amd -lpconv_coder -Wl,-rpath,$PWD/pseudocylindrical_convolution_amd
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pconv_coder.h"

namespace {
constexpr uint32_t kTop = 0x80000000u;

struct BitReader {  // big-endian bit stream, 64-bit window
  const uint8_t *p, *end;
  uint64_t acc;
  int have;
  __device__ void init(const uint8_t *data, size_t n) { p = data; end = data + n; acc = 0; have = 0; }
  __device__ uint32_t get(int n) {  // 0 < n <= 32; zeros past the end (as the reference's reader)
    if (have < n) {
      while (have <= 56) {
        acc |= (uint64_t)(p < end ? *p : 0) << (56 - have);
        p++;
        have += 8;
      }
    }
    const uint32_t v = (uint32_t)(acc >> (64 - n));
    acc <<= n;
    have -= n;
    return v;
  }
};

// one wave; every lane runs the same scalar state machine, lane k < 9 owns threshold k
__global__ __launch_bounds__(64) void ac_decode_kernel(const int32_t *__restrict__ rows, const uint8_t *__restrict__ bytes,
                                                       size_t nbytes, int32_t *__restrict__ out, int n) {
  const int lane = threadIdx.x;
  BitReader br;
  br.init(bytes, nbytes);
  uint32_t low = 0, high = 0xffffffffu, code = br.get(32);
  for (int i = 0; i < n; i++) {
    const uint64_t range = (uint64_t)high - low + 1;
    const uint32_t offset = code - low;
    const uint32_t t = (uint32_t)rows[(size_t)i * 9 + (lane < 9 ? lane : 8)];
    const uint64_t thr = ((uint64_t)t * range) >> 16;          // <= 2^32
    const unsigned long long le = __ballot(lane >= 1 && lane < 8 && thr <= offset);
    const int sym = __popcll(le);
    const uint64_t lo_thr = __shfl(thr, sym), hi_thr = __shfl(thr, sym + 1);
    const uint32_t base = low;
    low = base + (uint32_t)lo_thr;
    high = base + (uint32_t)(hi_thr - 1);
    const int agree = __clz((int)(low ^ high));
    if (agree > 0) {
      code = (agree == 32 ? 0u : code << agree) | br.get(agree);
      low = agree == 32 ? 0u : low << agree;
      high = agree == 32 ? 0xffffffffu : (high << agree) | ((1u << agree) - 1);
    }
    const uint32_t pattern = (low & ~high & 0x7fffffffu) << 1;
    const int squeeze = __clz((int)~pattern);
    if (squeeze > 0) {
      code = (code & kTop) | ((code << squeeze) & 0x7fffffffu) | br.get(squeeze);
      low = (low << squeeze) & 0x7fffffffu;
      high = ((high << squeeze) & 0x7fffffffu) | kTop | ((1u << squeeze) - 1);
    }
    if (lane == 0) out[i] = sym;
  }
}

// ---- the same decoder with the step's rows and byte window resident in LDS -------------------------
constexpr int kStep = 1920;          // symbols of one wavefront step of a 4096x2048 frame (one frame group)
constexpr int kWindow = 2 * kStep;   // bytes the step may consume: 16 bits per symbol is the coder's worst case

struct LdsBitReader {  // big-endian bit stream over an LDS window, 64-bit accumulator
  const uint8_t *w;    // LDS
  int pos, have;
  uint64_t acc;
  __device__ void init(const uint8_t *window) { w = window; pos = 0; have = 0; acc = 0; }
  __device__ uint32_t get(int n) {  // 0 < n <= 32
    if (have < n) {
      // refill 4 bytes at a time from LDS (byte-aligned positions: assemble from single bytes)
      while (have <= 32) {
        const uint32_t four = ((uint32_t)w[pos] << 24) | ((uint32_t)w[pos + 1] << 16) | ((uint32_t)w[pos + 2] << 8) | w[pos + 3];
        acc |= (uint64_t)four << (32 - have);
        pos += 4;
        have += 32;
      }
    }
    const uint32_t v = (uint32_t)(acc >> (64 - n));
    acc <<= n;
    have -= n;
    return v;
  }
};

__global__ __launch_bounds__(64) void ac_decode_lds_kernel(const int32_t *__restrict__ rows, const uint8_t *__restrict__ bytes,
                                                           size_t nbytes, int32_t *__restrict__ out, int n) {
  __shared__ int32_t row_s[kStep * 9];
  __shared__ uint8_t win_s[kWindow + 16];
  __shared__ int32_t sym_s[kStep];
  const int lane = threadIdx.x;
  size_t consumed_bits = 0;  // stream position at the start of the step
  uint32_t low = 0, high = 0xffffffffu, code = 0;
  bool primed = false;
  for (int s0 = 0; s0 < n; s0 += kStep) {
    const int len = n - s0 < kStep ? n - s0 : kStep;
    // all lanes: this step's rows and byte window -> LDS (what a fused table kernel would leave there)
    for (int e = lane; e < len * 9; e += 64) row_s[e] = rows[(size_t)s0 * 9 + e];
    const size_t byte0 = consumed_bits >> 3;
    for (int e = lane; e < kWindow + 16; e += 64) win_s[e] = byte0 + e < nbytes ? bytes[byte0 + e] : 0;
    __syncthreads();
    LdsBitReader br;
    br.init(win_s);
    if (consumed_bits & 7) (void)br.get((int)(consumed_bits & 7));
    size_t used = consumed_bits & 7;
    if (!primed) {
      code = br.get(32);
      used += 32;
      primed = true;
    }
    uint32_t t = (uint32_t)row_s[lane < 9 ? lane : 8];
    for (int i = 0; i < len; i++) {
      // the next symbol's row is requested before this symbol's dependent chain starts
      const uint32_t tnext = (uint32_t)row_s[(i + 1 < len ? i + 1 : i) * 9 + (lane < 9 ? lane : 8)];
      const uint64_t range = (uint64_t)high - low + 1;
      const uint32_t offset = code - low;
      const uint64_t thr = ((uint64_t)t * range) >> 16;
      const unsigned long long le = __ballot(lane >= 1 && lane < 8 && thr <= offset);
      const int sym = __popcll(le);
      const uint64_t lo_thr = __shfl(thr, sym), hi_thr = __shfl(thr, sym + 1);
      const uint32_t base = low;
      low = base + (uint32_t)lo_thr;
      high = base + (uint32_t)(hi_thr - 1);
      const int agree = __clz((int)(low ^ high));
      if (agree > 0) {
        code = (agree == 32 ? 0u : code << agree) | br.get(agree);
        used += agree;
        low = agree == 32 ? 0u : low << agree;
        high = agree == 32 ? 0xffffffffu : (high << agree) | ((1u << agree) - 1);
      }
      const uint32_t pattern = (low & ~high & 0x7fffffffu) << 1;
      const int squeeze = __clz((int)~pattern);
      if (squeeze > 0) {
        code = (code & kTop) | ((code << squeeze) & 0x7fffffffu) | br.get(squeeze);
        used += squeeze;
        low = (low << squeeze) & 0x7fffffffu;
        high = ((high << squeeze) & 0x7fffffffu) | kTop | ((1u << squeeze) - 1);
      }
      if (lane == 0) sym_s[i] = sym;
      t = tnext;
    }
    consumed_bits = (consumed_bits & ~(size_t)7) + used;
    __syncthreads();
    for (int e = lane; e < len; e += 64) out[s0 + e] = sym_s[e];  // (a fused scatter kernel would read them from LDS)
    __syncthreads();
  }
}
}  // namespace

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 400000;
  std::vector<int32_t> rows((size_t)n * 9), sym(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
  for (int i = 0; i < n; i++) {
    // a peaked 8-symbol distribution like the codec's (about 1.4 bits per symbol), every bin >= 1 count
    uint32_t cuts[7];
    const uint32_t centre = 8192 + rnd() % 49152, width = 2000 + rnd() % 20000;
    for (int k = 0; k < 7; k++) {
      const double z = (k - 3 + (rnd() % 1000) / 1000.0 - 0.5) * 0.9;
      double v = centre + z * width;
      v = v < 1 + k ? 1 + k : (v > 65535 - (7 - k) ? 65535 - (7 - k) : v);
      cuts[k] = (uint32_t)v;
    }
    rows[(size_t)i * 9] = 0;
    for (int k = 0; k < 7; k++) {
      uint32_t c = cuts[k];
      if (c <= (uint32_t)rows[(size_t)i * 9 + k]) c = rows[(size_t)i * 9 + k] + 1;
      rows[(size_t)i * 9 + k + 1] = (int32_t)c;
    }
    rows[(size_t)i * 9 + 8] = 65536;
    // draw the symbol from the row's own distribution
    const uint32_t u = rnd() % 65536;
    int k = 0;
    while (k < 7 && u >= (uint32_t)rows[(size_t)i * 9 + k + 1]) k++;
    sym[i] = k;
  }
  pconv_coder *enc = pconv_coder_new(nullptr);
  pconv_coder_start_encoder(enc);
  if (pconv_coder_encodes(enc, rows.data(), 8, sym.data(), n) < 0 || pconv_coder_end_encoder(enc) < 0) {
    printf("encode failed: %s\n", pconv_coder_error(enc));
    return 1;
  }
  size_t nbytes = 0;
  const uint8_t *bytes = pconv_coder_bytes(enc, &nbytes);
  printf("%d symbols, %zu bytes (%.3f bits/symbol)\n", n, nbytes, nbytes * 8.0 / n);

  // CPU decoder
  std::vector<int32_t> cpu(n);
  pconv_coder *dec = pconv_coder_new(nullptr);
  double best_cpu = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    pconv_coder_start_decoder_mem(dec, bytes, nbytes);
    const auto t0 = std::chrono::steady_clock::now();
    if (pconv_coder_decodes_i32(dec, rows.data(), 8, cpu.data(), n) < 0) {
      printf("cpu decode failed: %s\n", pconv_coder_error(dec));
      return 1;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    best_cpu = dt < best_cpu ? dt : best_cpu;
  }
  int bad = 0;
  for (int i = 0; i < n; i++) bad += cpu[i] != sym[i];
  printf("CPU decoder: %.1f ns/symbol, %d mismatches\n", best_cpu / n * 1e9, bad);

  // device decoder
  int32_t *rows_d, *out_d;
  uint8_t *bytes_d;
  hipMalloc(&rows_d, rows.size() * 4);
  hipMalloc(&out_d, (size_t)n * 4);
  hipMalloc(&bytes_d, nbytes + 16);
  hipMemcpy(rows_d, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bytes_d, bytes, nbytes, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<int32_t> gpu(n);
  int failed = 0;
  float per_kernel[2] = {0, 0};
  for (int which = 0; which < 2; which++) {
    float best = 1e9f;
    hipMemset(out_d, 0xff, (size_t)n * 4);
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      if (which == 0)
        hipLaunchKernelGGL(ac_decode_kernel, dim3(1), dim3(64), 0, 0, rows_d, bytes_d, nbytes, out_d, n);
      else
        hipLaunchKernelGGL(ac_decode_lds_kernel, dim3(1), dim3(64), 0, 0, rows_d, bytes_d, nbytes, out_d, n);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    hipMemcpy(gpu.data(), out_d, (size_t)n * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (int i = 0; i < n; i++) bad += gpu[i] != sym[i];
    failed += bad != 0;
    per_kernel[which] = best * 1e6f / n;
    printf("device decoder, one wave, %s: %.1f ns/symbol, %d mismatches\n",
           which == 0 ? "rows and bytes loaded from GLOBAL memory inside the serial loop (round 2's probe)"
                      : "step's rows and byte window resident in LDS, next row read ahead",
           per_kernel[which], bad);
  }
  printf("a decode step of one 4096x2048 frame has ~1920 symbols: %.1f us on the device (LDS-resident), %.1f us on a host core\n",
         per_kernel[1] * 1920 / 1e3, best_cpu / n * 1e9 * 1920 / 1e3);
  return failed != 0;
}
