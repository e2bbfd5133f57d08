// CPU arithmetic coder behind the `coder` module (libpconv_coder.so).
//
// Same bitstream as the reference's Nayuki-style coder (ArithmeticCoder.cpp:31-69,
// 72-170; BitIoStream.cpp:13-70), re-designed around word-level operations: the
// renormalisation loops that the reference runs one bit at a time (a virtual call
// and a stream put per bit) are resolved with count-leading-zeros, and bits are
// packed into a memory buffer that is written to disk once.  The interval
// arithmetic (uint64 products, truncating division by the table total) is kept
// operation for operation because it defines the stream.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/pconv_coder.h"

namespace {

constexpr int kStateBits = 32;
constexpr uint64_t kMaxRange = 1ull << kStateBits;
constexpr uint64_t kMinRange = (kMaxRange >> 2) + 2;
constexpr uint64_t kMaxTotal = kMinRange;  // min(UINT64_MAX / kMaxRange, kMinRange)
constexpr uint64_t kMask = kMaxRange - 1;
constexpr uint64_t kTop = kMaxRange >> 1;

inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

// x / total, exact: the codec's tables always sum to 65536, so the two 64-bit
// divisions per symbol of the textbook formulation become shifts
struct Total {
  uint32_t total;
  int shift;  // >= 0 when total is a power of two
  explicit Total(uint32_t t) : total(t), shift((t && !(t & (t - 1))) ? __builtin_ctz(t) : -1) {}
  inline uint64_t div(uint64_t x) const { return shift >= 0 ? (x >> shift) : x / total; }
};

class BitSink {
 public:
  void clear() {
    len_ = 0;
    acc_ = 0;
    fill_ = 0;
  }
  // room for `nbytes` more bytes (+ the accumulator's): lets a caller that knows an upper bound of what it is
  // going to put skip the per-put capacity check's reallocation path
  void reserve(size_t nbytes) {
    if (len_ + nbytes + 16 > bytes_.size()) bytes_.resize((len_ + nbytes + 16) * 2 + 4096);
  }
  // append the low `n` bits of v, most significant first (n <= 32): a 64-bit accumulator,
  // whole bytes leave it as soon as they are complete
  inline void put(uint64_t v, int n) {
    acc_ = (acc_ << n) | (v & ((1ull << n) - 1));
    fill_ += n;  // < 8 + 32
    if (len_ + 8 > bytes_.size()) bytes_.resize(bytes_.size() * 2 + 4096);
    while (fill_ >= 8) {
      fill_ -= 8;
      bytes_[len_++] = (uint8_t)(acc_ >> fill_);
    }
  }
  // the same for a caller that has reserve()d room for everything it is going to put: no capacity check, and
  // the accumulator is emptied four bytes at a time (it holds up to 31 + 32 bits), big-endian like put()
  inline void put_reserved(uint64_t v, int n) {
    acc_ = (acc_ << n) | (v & ((1ull << n) - 1));
    fill_ += n;
    if (fill_ >= 32) {
      fill_ -= 32;
      const uint32_t w = __builtin_bswap32((uint32_t)(acc_ >> fill_));
      memcpy(bytes_.data() + len_, &w, 4);
      len_ += 4;
    }
  }
  void put_run(int bit, uint64_t count) {
    const uint64_t pattern = bit ? 0xffffffffull : 0;
    while (count > 0) {
      int n = count > 32 ? 32 : (int)count;
      put(pattern, n);
      count -= n;
    }
  }
  void pad_to_byte() {
    put(0, (8 - fill_ % 8) % 8);  // (put() also empties the whole bytes put_reserved() may have left)
  }
  const uint8_t *data() const { return bytes_.data(); }
  size_t size() const { return len_; }

 private:
  std::vector<uint8_t> bytes_;
  size_t len_ = 0;
  uint64_t acc_ = 0;
  int fill_ = 0;
};

class BitSource {
 public:
  void reset(const uint8_t *p, size_t n) {
    p_ = p;
    n_ = n;
    pos_ = 0;
    acc_ = 0;
    have_ = 0;
  }
  // next n bits (n <= 32), zeros past the end (ArithmeticCoder.cpp:121-126).  The accumulator is refilled four
  // bytes at a time while the stream has them (one unaligned big-endian load instead of a byte loop).
  inline uint32_t get(int n) {
    if (have_ < n) {
      if (pos_ + 4 <= n_) {
        uint32_t w;
        memcpy(&w, p_ + pos_, 4);
        acc_ = (acc_ << 32) | __builtin_bswap32(w);
        pos_ += 4;
        have_ += 32;
      } else {
        while (have_ < n) {
          uint64_t b = pos_ < n_ ? p_[pos_] : 0;
          pos_++;
          acc_ = (acc_ << 8) | b;
          have_ += 8;
        }
      }
    }
    have_ -= n;
    return (uint32_t)((acc_ >> have_) & ((n == 32) ? 0xffffffffull : ((1ull << n) - 1)));
  }

 private:
  const uint8_t *p_ = nullptr;
  size_t n_ = 0, pos_ = 0;
  uint64_t acc_ = 0;  // (have_ <= 31 before a refill: 63 bits at most are live)
  int have_ = 0;
};

}  // namespace

struct pconv_coder {
  std::string path;
  bool has_path = false;
  std::string err;
  enum { IDLE, ENCODING, DECODING } mode = IDLE;
  uint64_t low = 0, high = kMask, code = 0;
  uint64_t pending = 0;  // deferred underflow bits (numUnderflow)
  BitSink sink;
  BitSource source;
  std::vector<uint8_t> file_bytes;

  int fail(int code_, const char *msg) {
    err = msg;
    return code_;
  }

  // narrow [low, high] to the symbol's sub-interval and renormalise
  // (ArithmeticCoderBase::update, ArithmeticCoder.cpp:31-69)
  template <bool kEncode>
  int narrow(const uint32_t *table, uint32_t total, uint32_t symbol) {
    if (low >= high || (low & kMask) != low || (high & kMask) != high)
      return fail(PCONV_CODER_ESTATE, "Assertion error: Low or high out of range");
    const uint64_t range = high - low + 1;
    if (range < kMinRange || range > kMaxRange)
      return fail(PCONV_CODER_ESTATE, "Assertion error: Range out of range");
    const uint32_t sym_low = table[symbol];
    const uint32_t sym_high = table[symbol + 1];
    if (sym_low == sym_high) return fail(PCONV_CODER_EZEROFREQ, "Symbol has zero frequency");
    if (total > kMaxTotal)
      return fail(PCONV_CODER_ETOTAL, "Cannot code symbol because total is too large");
    const Total tot(total);
    const uint64_t new_low = low + tot.div(sym_low * range);
    const uint64_t new_high = low + tot.div(sym_high * range) - 1;
    low = new_low;
    high = new_high;
    renormalise<kEncode>();
    return 0;
  }

  // shift out what low and high have settled (ArithmeticCoder.cpp:52-69 / 117-150)
  template <bool kEncode>
  inline void renormalise() {
    // leading bits on which low and high agree leave the state
    const int agree = clz32((uint32_t)((low ^ high) & kMask));
    if (agree > 0) {
      if (kEncode) {
        const uint32_t top = (uint32_t)(low >> (kStateBits - agree));
        const int first = (top >> (agree - 1)) & 1;
        sink.put(first, 1);
        if (pending) {
          sink.put_run(first ^ 1, pending);
          pending = 0;
        }
        if (agree > 1) sink.put(top & ((1u << (agree - 1)) - 1), agree - 1);
      } else {
        code = ((code << agree) & kMask) | source.get(agree);
      }
      low = (low << agree) & kMask;
      high = ((high << agree) & kMask) | ((1ull << agree) - 1);
    }
    // low = 01..., high = 10...: squeeze out the second-highest bit while it
    // keeps that pattern
    const uint32_t pattern = (uint32_t)((low & ~high & (kMask >> 1)) << 1);
    const int squeeze = clz32(~pattern);
    if (squeeze > 0) {
      if (kEncode) {
        pending += squeeze;
      } else {
        code = (code & kTop) | ((code << squeeze) & (kMask >> 1)) | source.get(squeeze);
      }
      low = (low << squeeze) & (kMask >> 1);
      high = ((high << squeeze) & (kMask >> 1)) | kTop | ((1ull << squeeze) - 1);
    }
  }

  // ArithmeticDecoder::read for the codec's tables: 8 symbols, total 65536.  The textbook form
  // divides ((offset + 1) * total - 1) by the range and binary-searches the table for the
  // quotient; with q = that quotient, table[k] <= q  <=>  table[k] * range < (offset + 1) * total
  // <=>  floor(table[k] * range / total) <= offset, so the symbol is the number of thresholds
  // floor(table[k] * range >> 16), k = 1..7, that do not exceed the offset: eight independent
  // multiplies and compares, no division, no data-dependent branch -- and the two thresholds
  // around the symbol are exactly the new interval bounds of update().  Same symbols, same state.
  inline int read_symbol_8x65536(const uint32_t *t) {
    const uint64_t range = high - low + 1;
    if (low >= high || range < kMinRange || range > kMaxRange)
      return fail(PCONV_CODER_ESTATE, "Assertion error: Range out of range");
    const uint64_t offset = code - low;
    uint64_t thr[9];
#pragma GCC unroll 9
    for (int k = 0; k < 9; k++) thr[k] = ((uint64_t)t[k] * range) >> 16;
    unsigned symbol = 0;
#pragma GCC unroll 7
    for (int k = 1; k < 8; k++) symbol += thr[k] <= offset;
    if (t[symbol] == t[symbol + 1]) return fail(PCONV_CODER_EZEROFREQ, "Symbol has zero frequency");
    if (offset < thr[symbol] || thr[symbol + 1] <= offset) return fail(PCONV_CODER_EDESYNC, "Assertion error");
    const uint64_t base = low;
    low = base + thr[symbol];
    high = base + thr[symbol + 1] - 1;
    renormalise<false>();
    if (code < low || code > high)
      return fail(PCONV_CODER_EDESYNC, "Assertion error: Code out of range");
    return (int)symbol;
  }

  // ArithmeticDecoder::read (ArithmeticCoder.cpp:82-115)
  int read_symbol(const uint32_t *table, uint32_t ncode, uint32_t total) {
    if (total > kMaxTotal)
      return fail(PCONV_CODER_ETOTAL, "Cannot decode symbol because total is too large");
    if (total == 0) return fail(PCONV_CODER_EARG, "table total is zero");
    const uint64_t range = high - low + 1;
    const uint64_t offset = code - low;
    const Total tot(total);
    const uint64_t value = ((offset + 1) * total - 1) / range;
    if (tot.div(value * range) > offset) return fail(PCONV_CODER_EDESYNC, "Assertion error");
    if (value >= total) return fail(PCONV_CODER_EDESYNC, "Assertion error");
    uint32_t start = 0, end = ncode;
    while (end - start > 1) {
      uint32_t middle = (start + end) >> 1;
      if (table[middle] > value)
        end = middle;
      else
        start = middle;
    }
    if (start + 1 != end) return fail(PCONV_CODER_EDESYNC, "Assertion error");
    const uint32_t symbol = start;
    if (offset < tot.div(table[symbol] * range) || tot.div(table[symbol + 1] * range) <= offset)
      return fail(PCONV_CODER_EDESYNC, "Assertion error");
    int rc = narrow<false>(table, total, symbol);
    if (rc < 0) return rc;
    if (code < low || code > high)
      return fail(PCONV_CODER_EDESYNC, "Assertion error: Code out of range");
    return (int)symbol;
  }
};

// The codec's rows (8 symbols, total 65536) in one loop with the coder's state -- interval, code word, bit
// reader -- in registers: read_symbol_8x65536 symbol by symbol keeps them in the object (every call reloads and
// stores them, and its error paths make the compiler assume they may alias).  Same operations, same order, same
// checks; rows of another shape end the fast loop (the caller goes on with the general path).  Returns the
// number of rows done, or < 0.
// Where the rows of the fast loops come from.  RowsI32: the reference's int32[n][9] tables.  Rows16: the engine's
// packed rows, 16 bytes per symbol -- uint16 c1 .. c7 (c0 = 0 and c8 = 65536 are implied) and an auxiliary word:
// bits 0-7 the symbol (encoder), bit 7 + k set when c_k is 65536 and was stored as 0 (a row whose top bins are
// empty: kept exact, so that the coder reports it exactly as it would for the int32 row).  Less than half the
// bytes of the int32 rows + labels cross PCIe (16 instead of 40 per symbol).
struct RowsI32 {
  const int32_t *table;
  const int32_t *symbols;
  inline const uint32_t *row(int i, uint32_t *) const { return reinterpret_cast<const uint32_t *>(table + (size_t)i * 9); }
  inline uint32_t symbol(int i) const { return (uint32_t)symbols[i]; }
};
struct Rows16 {
  const uint16_t *rows;
  inline const uint32_t *row(int i, uint32_t *t) const {
    const uint16_t *r = rows + (size_t)i * 8;
    const uint32_t aux = r[7];
    t[0] = 0;
    // (the widening of the seven entries is one vector instruction; the flag bits are all clear except in rows
    // whose top bins are empty, or that the GPU marked as not of this shape)
#pragma GCC unroll 7
    for (int k = 1; k < 8; k++) t[k] = (uint32_t)r[k - 1];
    t[8] = 65536u;
    if (__builtin_expect((aux & 0xff00u) != 0, 0)) {
#pragma GCC unroll 7
      for (int k = 1; k < 8; k++) t[k] += ((aux >> (7 + k)) & 1u) << 16;
      if (aux & 0x8000u) t[8] = 0;  // not a row of this shape: ends the fast loop, the caller reports it
    }
    return t;
  }
  inline uint32_t symbol(int i) const { return rows[(size_t)i * 8 + 7] & 0xffu; }
};

template <typename T, typename Rows>
static int decode_rows_8x65536(pconv_coder *c, const Rows &rows, T *out, int n) {
  uint64_t low = c->low, high = c->high, code = c->code;
  BitSource src = c->source;
  int i = 0, rc = 0;
  for (; i < n; i++) {
    uint32_t unpacked[9];
    const uint32_t *t = rows.row(i, unpacked);
    if (t[8] != 65536u) break;
    const uint64_t range = high - low + 1;
    if (low >= high || range < kMinRange || range > kMaxRange) {
      rc = c->fail(PCONV_CODER_ESTATE, "Assertion error: Range out of range");
      break;
    }
    const uint64_t offset = code - low;
    uint64_t thr[9];
#pragma GCC unroll 9
    for (int k = 0; k < 9; k++) thr[k] = ((uint64_t)t[k] * range) >> 16;
    unsigned symbol = 0;
#pragma GCC unroll 7
    for (int k = 1; k < 8; k++) symbol += thr[k] <= offset;
    const uint64_t lo_t = thr[symbol], hi_t = thr[symbol + 1];
    if (t[symbol] == t[symbol + 1]) {
      rc = c->fail(PCONV_CODER_EZEROFREQ, "Symbol has zero frequency");
      break;
    }
    if (offset < lo_t || hi_t <= offset) {
      rc = c->fail(PCONV_CODER_EDESYNC, "Assertion error");
      break;
    }
    high = low + hi_t - 1;
    low = low + lo_t;
    // renormalise<false>() on the local state
    const int agree = clz32((uint32_t)((low ^ high) & kMask));
    if (agree > 0) {
      code = ((code << agree) & kMask) | src.get(agree);
      low = (low << agree) & kMask;
      high = ((high << agree) & kMask) | ((1ull << agree) - 1);
    }
    const uint32_t pattern = (uint32_t)((low & ~high & (kMask >> 1)) << 1);
    const int squeeze = clz32(~pattern);
    if (squeeze > 0) {
      code = (code & kTop) | ((code << squeeze) & (kMask >> 1)) | src.get(squeeze);
      low = (low << squeeze) & (kMask >> 1);
      high = ((high << squeeze) & (kMask >> 1)) | kTop | ((1ull << squeeze) - 1);
    }
    if (code < low || code > high) {
      rc = c->fail(PCONV_CODER_EDESYNC, "Assertion error: Code out of range");
      break;
    }
    out[i] = (T)symbol;
  }
  c->low = low, c->high = high, c->code = code;
  c->source = src;
  return rc < 0 ? rc : i;
}

template <typename T>
static int decode_many(pconv_coder *c, const int32_t *table, int ncode, T *out, int n) {
  if (!c || ncode <= 0) return PCONV_CODER_EARG;
  if (c->mode != pconv_coder::DECODING) return c->fail(PCONV_CODER_ESTATE, "decoder not started");
  if (n <= 0) return 0;
  if (!table || !out) return c->fail(PCONV_CODER_EARG, "null table or output");
  const int stride = ncode + 1;
  int first = 0;
  if (ncode == 8) {
    first = decode_rows_8x65536(c, RowsI32{table, nullptr}, out, n);
    if (first < 0) return first;
  }
  for (int i = first; i < n; i++) {
    const uint32_t *row = reinterpret_cast<const uint32_t *>(table + (size_t)i * stride);
    const int s = (ncode == 8 && row[8] == 65536u) ? c->read_symbol_8x65536(row)
                                                   : c->read_symbol(row, (uint32_t)ncode, row[ncode]);
    if (s < 0) return s;
    out[i] = (T)s;
  }
  return 0;
}

// The codec's rows (8 symbols, total 65536) in a row: returns the number of rows coded (rows of another shape end
// the loop: the caller goes on with the general path), or < 0.
template <typename Rows>
static int encode_rows_8x65536(pconv_coder *c, const Rows &rows, int n) {
  int i = 0;
  {
    // the codec's rows (8 symbols, total 65536): narrow<true>() with the interval in registers and the two
    // divisions by the total as shifts; rows of another shape end the fast loop
    uint64_t low = c->low, high = c->high, pending = c->pending;
    // <= 33 bits leave the state per symbol, deferred bits of THIS call included; the bits deferred by earlier
    // calls (`pending` at entry) are flushed through the checked put_run() but land in the same buffer
    c->sink.reserve((size_t)n * 5 + 64 + (size_t)(pending / 8) + 8);
    int rc = 0;
    for (; i < n; i++) {
      uint32_t unpacked[9];
      const uint32_t *row = rows.row(i, unpacked);
      if (row[8] != 65536u) break;
      const uint32_t s = rows.symbol(i);
      if (s >= 8u) {
        rc = c->fail(PCONV_CODER_EARG, "symbol out of range");
        break;
      }
      const uint64_t range = high - low + 1;
      if (low >= high || range < kMinRange || range > kMaxRange) {
        rc = c->fail(PCONV_CODER_ESTATE, "Assertion error: Range out of range");
        break;
      }
      const uint32_t sym_low = row[s], sym_high = row[s + 1];
      if (sym_low == sym_high) {
        rc = c->fail(PCONV_CODER_EZEROFREQ, "Symbol has zero frequency");
        break;
      }
      high = low + ((sym_high * range) >> 16) - 1;
      low = low + ((sym_low * range) >> 16);
      const int agree = clz32((uint32_t)((low ^ high) & kMask));
      if (agree > 0) {
        const uint32_t top = (uint32_t)(low >> (kStateBits - agree));
        if (__builtin_expect(pending == 0, 1)) {
          c->sink.put_reserved(top, agree);  // (the first bit and the rest in one piece)
        } else {
          const int firstbit = (top >> (agree - 1)) & 1;
          c->sink.put(firstbit, 1);
          c->sink.put_run(firstbit ^ 1, pending);
          pending = 0;
          if (agree > 1) c->sink.put(top & ((1u << (agree - 1)) - 1), agree - 1);
        }
        low = (low << agree) & kMask;
        high = ((high << agree) & kMask) | ((1ull << agree) - 1);
      }
      const uint32_t pattern = (uint32_t)((low & ~high & (kMask >> 1)) << 1);
      const int squeeze = clz32(~pattern);
      if (squeeze > 0) {
        pending += squeeze;
        low = (low << squeeze) & (kMask >> 1);
        high = ((high << squeeze) & (kMask >> 1)) | kTop | ((1ull << squeeze) - 1);
      }
    }
    c->low = low, c->high = high, c->pending = pending;
    if (rc < 0) return rc;
  }
  return i;
}

extern "C" {

pconv_coder *pconv_coder_new(const char *path) {
  pconv_coder *c = new pconv_coder();
  if (path) {
    c->path = path;
    c->has_path = true;
  }
  return c;
}

void pconv_coder_free(pconv_coder *c) { delete c; }

const char *pconv_coder_error(const pconv_coder *c) { return c ? c->err.c_str() : "null coder"; }

int pconv_coder_start_encoder(pconv_coder *c) {
  if (!c) return PCONV_CODER_EARG;
  if (c->has_path) {
    // the reference opens (truncates) the file here (coder.h:13-15)
    FILE *f = fopen(c->path.c_str(), "wb");
    if (!f) return c->fail(PCONV_CODER_EIO, "cannot open output file");
    fclose(f);
  }
  c->sink.clear();
  c->low = 0;
  c->high = kMask;
  c->pending = 0;
  c->mode = pconv_coder::ENCODING;
  return 0;
}

int pconv_coder_encode(pconv_coder *c, const uint32_t *table, uint32_t ncode, uint32_t sum,
                       uint32_t symbol) {
  if (!c || !table) return PCONV_CODER_EARG;
  if (c->mode != pconv_coder::ENCODING) return c->fail(PCONV_CODER_ESTATE, "encoder not started");
  if (symbol >= ncode) return c->fail(PCONV_CODER_EARG, "symbol out of range");
  return c->narrow<true>(table, sum, symbol);
}

int pconv_coder_encodes(pconv_coder *c, const int32_t *table, int ncode, const int32_t *symbols,
                        int n) {
  if (!c || ncode <= 0) return PCONV_CODER_EARG;
  if (c->mode != pconv_coder::ENCODING) return c->fail(PCONV_CODER_ESTATE, "encoder not started");
  if (n <= 0) return 0;
  if (!table || !symbols) return c->fail(PCONV_CODER_EARG, "null table or symbols");
  const int stride = ncode + 1;
  int i = 0;
  if (ncode == 8) {
    i = encode_rows_8x65536(c, RowsI32{table, symbols}, n);
    if (i < 0) return i;
  }
  for (; i < n; i++) {
    const uint32_t *row = reinterpret_cast<const uint32_t *>(table + (size_t)i * stride);
    const uint32_t s = (uint32_t)symbols[i];
    if (s >= (uint32_t)ncode) return c->fail(PCONV_CODER_EARG, "symbol out of range");
    int rc = c->narrow<true>(row, row[ncode], s);
    if (rc < 0) return rc;
  }
  return 0;
}

int pconv_coder_end_encoder(pconv_coder *c) {
  if (!c) return PCONV_CODER_EARG;
  if (c->mode != pconv_coder::ENCODING) return c->fail(PCONV_CODER_ESTATE, "encoder not started");
  c->sink.put(1, 1);  // ArithmeticEncoder::finish
  c->sink.pad_to_byte();
  c->mode = pconv_coder::IDLE;
  if (c->has_path) {
    FILE *f = fopen(c->path.c_str(), "wb");
    if (!f) return c->fail(PCONV_CODER_EIO, "cannot open output file");
    const size_t nb = c->sink.size();
    size_t wr = nb ? fwrite(c->sink.data(), 1, nb, f) : 0;
    fclose(f);
    if (wr != nb) return c->fail(PCONV_CODER_EIO, "short write");
  }
  return 0;
}

const uint8_t *pconv_coder_bytes(const pconv_coder *c, size_t *nbytes) {
  if (!c) return nullptr;
  if (nbytes) *nbytes = c->sink.size();
  return c->sink.data();
}

static int begin_decode(pconv_coder *c, const uint8_t *p, size_t n) {
  c->source.reset(p, n);
  c->low = 0;
  c->high = kMask;
  c->code = c->source.get(kStateBits);
  c->mode = pconv_coder::DECODING;
  return 0;
}

int pconv_coder_start_decoder(pconv_coder *c) {
  if (!c) return PCONV_CODER_EARG;
  if (!c->has_path) return c->fail(PCONV_CODER_EIO, "coder has no file path");
  FILE *f = fopen(c->path.c_str(), "rb");
  c->file_bytes.clear();
  if (f) {  // a missing file reads as an empty stream, like an unopened ifstream
    uint8_t buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0)
      c->file_bytes.insert(c->file_bytes.end(), buf, buf + got);
    fclose(f);
  }
  return begin_decode(c, c->file_bytes.data(), c->file_bytes.size());
}

int pconv_coder_start_decoder_mem(pconv_coder *c, const uint8_t *data, size_t nbytes) {
  if (!c || (!data && nbytes)) return PCONV_CODER_EARG;
  return begin_decode(c, data, nbytes);
}

int pconv_coder_decode(pconv_coder *c, const uint32_t *table, uint32_t ncode, uint32_t sum) {
  if (!c || !table) return PCONV_CODER_EARG;
  if (c->mode != pconv_coder::DECODING) return c->fail(PCONV_CODER_ESTATE, "decoder not started");
  return c->read_symbol(table, ncode, sum);
}

int pconv_coder_decodes(pconv_coder *c, const int32_t *table, int ncode, float *out, int n) {
  return decode_many<float>(c, table, ncode, out, n);
}

int pconv_coder_decodes_i32(pconv_coder *c, const int32_t *table, int ncode, int32_t *out, int n) {
  return decode_many<int32_t>(c, table, ncode, out, n);
}

int pconv_coder_encodes_rows16(pconv_coder *c, const uint16_t *rows, int n) {
  if (!c) return PCONV_CODER_EARG;
  if (c->mode != pconv_coder::ENCODING) return c->fail(PCONV_CODER_ESTATE, "encoder not started");
  if (n <= 0) return 0;
  if (!rows) return c->fail(PCONV_CODER_EARG, "null rows");
  const int done = encode_rows_8x65536(c, Rows16{rows}, n);
  if (done < 0) return done;
  return done == n ? 0 : c->fail(PCONV_CODER_EARG, "packed rows are 8 symbols of total 65536");
}

int pconv_coder_decodes_rows16_i32(pconv_coder *c, const uint16_t *rows, int32_t *out, int n) {
  if (!c) return PCONV_CODER_EARG;
  if (c->mode != pconv_coder::DECODING) return c->fail(PCONV_CODER_ESTATE, "decoder not started");
  if (n <= 0) return 0;
  if (!rows || !out) return c->fail(PCONV_CODER_EARG, "null rows or output");
  const int done = decode_rows_8x65536(c, Rows16{rows}, out, n);
  if (done < 0) return done;
  return done == n ? 0 : c->fail(PCONV_CODER_EARG, "packed rows are 8 symbols of total 65536");
}

}  // extern "C"
