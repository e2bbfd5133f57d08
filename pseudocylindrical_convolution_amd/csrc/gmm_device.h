// GMM parameters -> integer CDF row, shared by the per-op kernels (entropy.hip)
// and the engine kernels (entropy_engine.hip) so both produce identical tables.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pconv_detmath.h"

// ---- GMM -> integer CDF row (entropy_gmm_table_cuda.cu:29-57,83-105,136-153) ----
constexpr int kMaxGauss = 16;

// softmax over the mixture weights and delta = max(delta, 0) + beta, on registers
__device__ __forceinline__ void gmm_prepare_row(float *wt, float *dl, int ng, float beta) {
  float mval = -1e10, psum = 0;
  for (int k = 0; k < ng; k++)
    if (mval < wt[k]) mval = wt[k];
  for (int k = 0; k < ng; k++) {
    wt[k] = pconv_expf(wt[k] - mval);
    psum += wt[k];
  }
  for (int k = 0; k < ng; k++) {
    wt[k] = wt[k] / psum;
    dl[k] = dl[k] < 0 ? beta : dl[k] + beta;
  }
}

// raw entry pt (1..nstep) of the integer CDF, before the monotonicity repair: exactly the
// value `cur` of gmm_cdf_row below at that pt (batch arithmetic, :148)
__device__ __forceinline__ float gmm_cdf_entry(const float *wt, const float *dl, const float *mu, int ng, int nstep,
                                               int pt, float bias, float total) {
  if (pt == nstep) return static_cast<int>(total);
  const float s2 = 1. / sqrt(2.0);
  float v = pt - 1 - bias + 0.5, ps = 0;
  for (int k = 0; k < ng; k++) {
    const float e = pconv_erff(s2 * (v - mu[k]) / dl[k]);
    ps = ps + wt[k] * (0.5 + 0.5 * e);  // double inside, as :148
  }
  return static_cast<int>(total * ps + 0.5);
}

// row[0..nstep]: integer CDF with the reference's monotonicity repair applied on
// the fly (every bin at least one count, taken back from the widest bin)
template <typename Out>
__device__ __forceinline__ void gmm_cdf_row(const float *wt, const float *dl, const float *mu, int ng,
                                            int nstep, float bias, float total, int batch_arith,
                                            Out *row) {
  const float s2 = 1. / sqrt(2.0);
  float prev = 0.f, shift = 0.f, widest = 0.f;
  int widest_at = 0;
  row[0] = (Out)0;
  for (int pt = 1; pt <= nstep; pt++) {
    float cur;
    if (pt == nstep) {
      cur = static_cast<int>(total);
    } else {
      float v = pt - 1 - bias + 0.5, ps = 0;
      for (int k = 0; k < ng; k++) {
        const float e = pconv_erff(s2 * (v - mu[k]) / dl[k]);
        if (batch_arith) {
          ps = ps + wt[k] * (0.5 + 0.5 * e);  // double inside, as :148
        } else {
          const float f = 0.5 + 0.5 * e;  // rounded to float, as :72-73
          ps = ps + wt[k] * f;
        }
      }
      cur = static_cast<int>(total * ps + 0.5);
    }
    // check kernel: compares the raw entry with the already shifted previous one
    if (cur <= prev) shift += 1;
    cur += shift;
    if (cur - prev > widest) {
      widest = cur - prev;
      widest_at = pt - 1;
    }
    row[pt] = (Out)cur;
    prev = cur;
  }
  if (shift > 0)
    for (int pt = widest_at; pt < nstep; pt++) row[pt + 1] = (Out)((float)row[pt + 1] - shift);
}

