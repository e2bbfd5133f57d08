// TEST INFRASTRUCTURE -- not part of the product.
// extern "C" wrapper around the REFERENCE arithmetic coder, compiled from the
// reference's own sources where they lie (/root/reference/coder/*.cpp; see
// oracle/Makefile).  Output: oracle/_ref/libcoder_ref.so (git-ignored).  Used to
// pin oracle/ac_oracle.py and the product coder against the real thing and to
// generate tests/golden/coder_*.npz (tools/gen_golden_coder.py).
#include <cstdint>
#include <string>
#include "coder.h"  // -I/root/reference/coder

extern "C" {
void *refcoder_new(const char *path) { return new Coder(std::string(path)); }
void refcoder_free(void *c) { delete static_cast<Coder *>(c); }
void refcoder_start_encoder(void *c) { static_cast<Coder *>(c)->start_encoder(); }
void refcoder_end_encoder(void *c) { static_cast<Coder *>(c)->end_encoder(); }
void refcoder_start_decoder(void *c) { static_cast<Coder *>(c)->start_decoder(); }
// returns 0, or -1 when the reference throws
int refcoder_encodes(void *c, const int32_t *table, int ncode, const int32_t *sym, int n) {
  try {
    for (int i = 0; i < n; i++) {
      const uint32_t *row = reinterpret_cast<const uint32_t *>(table + (size_t)i * (ncode + 1));
      static_cast<Coder *>(c)->encode(row, (uint32_t)ncode, row[ncode], (uint32_t)sym[i]);
    }
  } catch (const char *) {
    return -1;
  }
  return 0;
}
int refcoder_decodes(void *c, const int32_t *table, int ncode, int32_t *out, int n) {
  try {
    for (int i = 0; i < n; i++) {
      const uint32_t *row = reinterpret_cast<const uint32_t *>(table + (size_t)i * (ncode + 1));
      out[i] = (int32_t) static_cast<Coder *>(c)->decode(row, (uint32_t)ncode, row[ncode]);
    }
  } catch (const char *) {
    return -1;
  }
  return 0;
}
}
