// Shared helpers of libpconv_hip.so: error reporting and launch geometry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pconv_hip.h"

void pconv_set_error(const char *fmt, ...);

#define PCONV_REQUIRE(cond, ...)     \
  do {                               \
    if (!(cond)) {                   \
      pconv_set_error(__VA_ARGS__);  \
      return PCONV_EINVAL;           \
    }                                \
  } while (0)

#define PCONV_LAUNCH_CHECK(name)                                          \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      pconv_set_error("%s: %s", name, hipGetErrorString(e__));            \
      return PCONV_ELAUNCH;                                               \
    }                                                                     \
  } while (0)

// MI355X: 256 CUs.  Memory-bound grid-stride kernels are capped at 8 blocks of
// 256 threads per CU so the grid stays resident and the tail is short.
static inline unsigned pconv_grid(long long work_items, int block = 256) {
  long long blocks = (work_items + block - 1) / block;
  const long long cap = 256LL * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }
