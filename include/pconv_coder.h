/*
 * pconv_coder.h -- C ABI of libpconv_coder.so, the CPU arithmetic coder.
 *
 * Drop-in boundary for the reference's native module `coder`
 * (reference: coder/python.cpp:63-73 binds class Coder of coder/coder.h:8-59).
 * The bitstream is the reference's: 32-bit-state arithmetic coding with
 * cumulative-frequency tables passed per symbol (ArithmeticCoder.cpp:31-69),
 * big-endian bit packing, one terminating 1 bit, zero padding to a byte
 * (ArithmeticCoder.cpp:144-147, BitIoStream.cpp:49-70); raw bytes, no header.
 *
 * Error convention: functions return 0 (or the decoded symbol) on success and a
 * negative PCONV_CODER_E* value where the reference throws a `const char*`
 * (ArithmeticCoder.cpp:17,37-50,93-115); pconv_coder_error() gives the text.
 */
#ifndef PCONV_CODER_H
#define PCONV_CODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCONV_CODER_ESTATE (-1)   /* low/high/range invariant violated, wrong mode */
#define PCONV_CODER_EZEROFREQ (-2) /* "Symbol has zero frequency"                 */
#define PCONV_CODER_ETOTAL (-3)   /* total larger than MAX_TOTAL                  */
#define PCONV_CODER_EIO (-4)      /* cannot open / write the file                 */
#define PCONV_CODER_EARG (-5)
#define PCONV_CODER_EDESYNC (-6)  /* decoder assertion (corrupt stream / tables)  */

typedef struct pconv_coder pconv_coder;

/* Coder(name)  (coder.h:58-61).  path may be NULL for a memory-only coder. */
pconv_coder *pconv_coder_new(const char *path);
void pconv_coder_free(pconv_coder *c);
const char *pconv_coder_error(const pconv_coder *c);

/* start_encoder / encode / end_encoder  (coder.h:12-27) */
int pconv_coder_start_encoder(pconv_coder *c);
int pconv_coder_encode(pconv_coder *c, const uint32_t *table, uint32_t ncode, uint32_t sum,
                       uint32_t symbol);
/* encodes(table int32[n, ncode+1], ncode, symbols int32[n], n)  (python.cpp:22-40);
 * the total of row i is table[i][ncode]. */
int pconv_coder_encodes(pconv_coder *c, const int32_t *table, int ncode, const int32_t *symbols,
                        int n);
/* writes the file when the coder has a path; the bytes stay available through
 * pconv_coder_bytes() until the next start_encoder. */
int pconv_coder_end_encoder(pconv_coder *c);
const uint8_t *pconv_coder_bytes(const pconv_coder *c, size_t *nbytes);

/* start_decoder / decode  (coder.h:28-37).  start_decoder reads the file;
 * start_decoder_mem decodes a caller-owned buffer (kept by reference). */
int pconv_coder_start_decoder(pconv_coder *c);
int pconv_coder_start_decoder_mem(pconv_coder *c, const uint8_t *data, size_t nbytes);
int pconv_coder_decode(pconv_coder *c, const uint32_t *table, uint32_t ncode, uint32_t sum);
/* decodes(table int32[n, ncode+1], ncode, n) -> float32[n]  (python.cpp:41-61) */
int pconv_coder_decodes(pconv_coder *c, const int32_t *table, int ncode, float *out, int n);
/* same, integer output (used by the native decode loop) */
int pconv_coder_decodes_i32(pconv_coder *c, const int32_t *table, int ncode, int32_t *out, int n);


/* The same two loops over PACKED rows of the codec's shape (8 symbols, total 65536): 16 bytes per symbol instead
 * of the 36 + 4 of an int32[9] row and its int32 label -- what the native entropy engine moves across PCIe
 * (csrc/engine.cpp; encodes / decodes of python.cpp:22-61 are called with exactly such rows by
 * pseudo_codec.py:107,153).  Row i = uint16 rows[8 i .. 8 i + 7]: c1 .. c7 of the CDF (c0 = 0 and c8 = 65536 are
 * implied), then an auxiliary word: bits 0-7 the symbol to encode (ignored by the decoder), bit 7 + k set when
 * c_k == 65536 (stored as 0).  Same interval arithmetic, same streams, same error returns as the int32 rows. */
int pconv_coder_encodes_rows16(pconv_coder *c, const uint16_t *rows, int n);
int pconv_coder_decodes_rows16_i32(pconv_coder *c, const uint16_t *rows, int32_t *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* PCONV_CODER_H */
