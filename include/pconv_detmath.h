/*
 * pconv_detmath.h -- deterministic expf / erff shared by the HIP kernels and by
 * any conforming CPU decoder.
 *
 * The integer CDF tables that drive the arithmetic coder are int(65536*p+0.5) of
 * fp32 math (reference: entropy_gmm_table_cuda.cu:136-153).  The reference calls
 * the CUDA runtime's erf/exp, so its tables are only reproducible on the same
 * CUDA build.  Here both functions are fixed sequences of IEEE fmaf / add / mul
 * (no libm, no contraction), so a stream written on the GPU decodes on any IEEE
 * machine.  Accuracy: |erff - erf| < 1.2e-7 absolute, expf within 2 ulp.
 *
 * Usable from C, C++ and HIP device code.
 */
#ifndef PCONV_DETMATH_H
#define PCONV_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PCONV_DM_FN static __host__ __device__ inline
#define PCONV_DM_TABLE static __device__ __constant__ const
#define PCONV_DM_HAVE_DEVICE 1
#else
#define PCONV_DM_FN static inline
#define PCONV_DM_TABLE static const
#endif

#include "pconv_detmath_tables.h"

#if defined(PCONV_DM_HAVE_DEVICE)
/* host mirror of the table for host-side callers inside the HIP library */
#undef PCONV_DM_TABLE
#define PCONV_DM_TABLE static const
#define pconv_erf_coef pconv_erf_coef_host
#include "pconv_detmath_tables.h"
#undef pconv_erf_coef
#endif

PCONV_DM_FN float pconv_dm_bits2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* 2^n for n in [-126, 127] */
PCONV_DM_FN float pconv_dm_pow2(int n) { return pconv_dm_bits2f((uint32_t)(n + 127) << 23); }

PCONV_DM_FN float pconv_expf(float x) {
  if (!(x == x)) return x;
  if (x > 88.7228394f) return pconv_dm_bits2f(0x7f800000u);
  if (x < -103.0f) return 0.0f;
  /* n = round(x / ln2); r = x - n*ln2 (Cody-Waite, two-constant split) */
  float n = floorf(fmaf(x, 1.44269502f, 0.5f));
  float r = fmaf(n, -0.693145752f, x);
  r = fmaf(n, -1.42860677e-06f, r);
  /* e^r, |r| <= 0.3467: degree-7 Taylor, Horner with fmaf */
  float p = 1.98412701e-04f;
  p = fmaf(p, r, 1.38888892e-03f);
  p = fmaf(p, r, 8.33333377e-03f);
  p = fmaf(p, r, 4.16666679e-02f);
  p = fmaf(p, r, 1.66666672e-01f);
  p = fmaf(p, r, 0.5f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  /* scale by 2^n in two steps so that neither factor leaves the normal range */
  int ni = (int)n;
  int h = ni / 2;
  return p * pconv_dm_pow2(h) * pconv_dm_pow2(ni - h);
}

PCONV_DM_FN float pconv_erff(float x) {
  if (!(x == x)) return x;
  float a = fabsf(x);
  float y;
  if (a >= 4.0f) {
    y = 1.0f;
  } else {
    int i = (int)(a * 4.0f);
    float h = a - (0.25f * (float)i + 0.125f);
#if defined(__HIP_DEVICE_COMPILE__)
    const float *c = pconv_erf_coef[i];
#elif defined(PCONV_DM_HAVE_DEVICE)
    const float *c = pconv_erf_coef_host[i];
#else
    const float *c = pconv_erf_coef[i];
#endif
    y = c[PCONV_ERF_DEG];
    for (int k = PCONV_ERF_DEG - 1; k >= 0; k--) y = fmaf(y, h, c[k]);
    if (y > 1.0f) y = 1.0f;
  }
  return x < 0 ? -y : y;
}

#endif /* PCONV_DETMATH_H */
