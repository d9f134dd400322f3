// ed_fit_dev.hpp -- device arithmetic of the per-sample beta-binomial fit (kernel group K5).
//
// The reference fits  cbind(test, reference) ~ 1  with aod::betabin (reference R/class_definition.R:118),
// a third-party package that is not in the reference tree: it maximises, by Nelder-Mead,
//     l(p, phi) = sum_e [ lchoose(n_e, y_e) + lbeta(a + y_e, b + n_e - y_e) - lbeta(a, b) ],
//     a = p (1 - phi)/phi,  b = (1 - p)(1 - phi)/phi.
// This file is a NEW algorithm for the same maximum-likelihood estimate: Newton's method on
// (eta, lambda) = (logit p, log(a + b)) with the exact gradient and Hessian, which need the digamma
// and trigamma functions at a + y, b + n - y and a + b + n for every exon.  Nothing here has a
// bit-level counterpart in the reference; the checker (oracle/) holds an independent CPU fit and the
// in-tree digamma/trigamma of the reference (src/VP_psi.c:409-485, :744-787) to compare against.
#pragma once
#include <hip/hip_runtime.h>
#include "ed_pmath.h"

namespace edfit {

// Horner step of the fit's series.  ed_pm_fma_k (ed_pmath.h) pins its coefficient to a fixed scalar register pair through inline assembly -- the strict
// emission kernels' register budget is built on that; ED_FIT_PLAIN_FMA (diagnostic builds) leaves the coefficient to the compiler here.
#ifdef ED_FIT_PLAIN_FMA
#define EDFIT_FMA_K(a, b, k) __builtin_fma((a), (b), (k))
#else
#define EDFIT_FMA_K(a, b, k) ed_pm_fma_k((a), (b), (k))
#endif

// Reciprocal and logarithm for the fit.  Nothing here has to match the checker bit for bit (the fit is
// compared by tolerance), so the reciprocal is v_rcp_f64 + two Newton steps (~1 ulp) instead of an IEEE
// division, and the logarithm is ed_plog's algorithm on top of that reciprocal.
__device__ __forceinline__ double frcp(double b)
{
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  return __builtin_fma(r, e, r);
}

__device__ __forceinline__ double flog(double x)   // x normal, positive, finite
{
  const double c[ED_PM_LOG_NC] = ED_PM_LOG_COEFFS;
  const uint64_t u = ed_pm_bits(x);
  int k = (int)(u >> 52) - 1023;
  double m = ed_pm_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > ED_PM_SQRT2) { m = m * 0.5; k += 1; }
  const double f = m - 1.0;
  const double s = f * frcp(2.0 + f);
  const double z = s * s;
  double g = c[ED_PM_LOG_NC - 1];
#pragma unroll
  for (int i = ED_PM_LOG_NC - 2; i >= 0; --i) g = EDFIT_FMA_K(g, z, c[i]);
  const double hfsq = (0.5 * f) * f;
  const double dk = (double)k;
  const double w = __builtin_fma(s, hfsq + z * g, dk * ED_PM_LN2_LO);
  return __builtin_fma(dk, ED_PM_LN2_HI, f - (hfsq - w));
}

// psi(x) and psi'(x) for x > 0: upward recurrence to x >= 10, then the asymptotic (Stirling) series
// with 7 Bernoulli terms (truncation < 2e-16 relative at x = 10).  psi_nolog leaves the ln x term to
// the caller, who can merge the logarithms of several arguments into one (ln a - ln b = ln(a/b)).
__device__ __forceinline__ void digamma_trigamma_nolog(double x, double& xs, double& psi_rest, double& psi1)
{
  double s0 = 0.0, s1 = 0.0;
  while (x < 10.0) {   // divergent only for small shape parameters (high dispersion, low counts)
    const double r = frcp(x);
    s0 += r;
    s1 = __builtin_fma(r, r, s1);
    x += 1.0;
  }
  const double r = frcp(x);
  const double w = r * r;
  // psi(x)  = ln x - 1/(2x) - sum_k B_2k / (2k x^2k)
  double p = 1.0 / 12.0;                       // B14/14
  p = EDFIT_FMA_K(p, w, -691.0 / 32760.0);   // B12/12
  p = EDFIT_FMA_K(p, w, 1.0 / 132.0);        // B10/10
  p = EDFIT_FMA_K(p, w, -1.0 / 240.0);       // B8/8
  p = EDFIT_FMA_K(p, w, 1.0 / 252.0);        // B6/6
  p = EDFIT_FMA_K(p, w, -1.0 / 120.0);       // B4/4
  p = EDFIT_FMA_K(p, w, 1.0 / 12.0);         // B2/2
  psi_rest = (-0.5 * r - p * w) - s0;          // psi(x_original) = ln(xs) + psi_rest
  xs = x;
  // psi'(x) = 1/x + 1/(2x^2) + sum_k B_2k / x^(2k+1)
  double q = 7.0 / 6.0;                        // B14
  q = EDFIT_FMA_K(q, w, -691.0 / 2730.0);    // B12
  q = EDFIT_FMA_K(q, w, 5.0 / 66.0);         // B10
  q = EDFIT_FMA_K(q, w, -1.0 / 30.0);        // B8
  q = EDFIT_FMA_K(q, w, 1.0 / 42.0);         // B6
  q = EDFIT_FMA_K(q, w, -1.0 / 30.0);        // B4
  q = EDFIT_FMA_K(q, w, 1.0 / 6.0);          // B2
  psi1 = __builtin_fma(q * w, r, __builtin_fma(0.5, w, r)) + s1;
}

__device__ __forceinline__ void digamma_trigamma(double x, double& psi, double& psi1)
{
  double xs, rest;
  digamma_trigamma_nolog(x, xs, rest, psi1);
  psi = flog(xs) + rest;
}

// accumulators of one sample: gradient and Hessian of the log-likelihood with respect to (a, b),
// without the per-sample constant terms (those are added once per sample in the update kernel)
struct Acc {
  double ga, gb, haa, hab, hbb;
};

__device__ __forceinline__ void accumulate_cell(Acc& acc, double a, double b, double th, int y, int n)
{
  if (n <= 0) return;   // lbeta(a+0, b+0) - lbeta(a, b) = 0: the cell carries no information
  double x1, r1, q1, x2, r2, q2, x3, r3, q3;
  digamma_trigamma_nolog(a + (double)y, x1, r1, q1);
  digamma_trigamma_nolog(b + (double)(n - y), x2, r2, q2);
  digamma_trigamma_nolog(th + (double)n, x3, r3, q3);
  // psi(x1) - psi(x3) = ln(x1/x3) + (r1 - r3): two logarithms per cell instead of three
  const double i3 = frcp(x3);
  acc.ga += flog(x1 * i3) + (r1 - r3);
  acc.gb += flog(x2 * i3) + (r2 - r3);
  acc.haa += q1 - q3;
  acc.hab -= q3;
  acc.hbb += q2 - q3;
}

// psi / psi' without the logarithm for an argument known to be >= 32: four Bernoulli terms (the fifth is below 7e-18 there)
__device__ __forceinline__ void digamma_trigamma_nolog_big(double x, double& psi_rest, double& psi1)
{
  const double r = frcp(x);
  const double w = r * r;
  double p = -1.0 / 240.0;                     // B8/8
  p = EDFIT_FMA_K(p, w, 1.0 / 252.0);        // B6/6
  p = EDFIT_FMA_K(p, w, -1.0 / 120.0);       // B4/4
  p = EDFIT_FMA_K(p, w, 1.0 / 12.0);         // B2/2
  psi_rest = -0.5 * r - p * w;
  double q = -1.0 / 30.0;                      // B8
  q = EDFIT_FMA_K(q, w, 1.0 / 42.0);         // B6
  q = EDFIT_FMA_K(q, w, -1.0 / 30.0);        // B4
  q = EDFIT_FMA_K(q, w, 1.0 / 6.0);          // B2
  psi1 = __builtin_fma(q * w, r, __builtin_fma(0.5, w, r));
}

// accumulate_cell for a run of cells (k_fit_accum).  The gradient needs sum_e ln(x1 / x3) and sum_e ln(x2 / x3): the ratios of a run are MULTIPLIED
// (pa, pb) and the caller takes one logarithm per run of eight cells instead of two per cell -- ln of a product of eight factors carries the same
// ~1e-15 of absolute error as the sum of eight rounded logarithms, and the factors (>= 10 / 2^31 each: the arguments come back shifted to >= 10)
// cannot leave the range.  `big`: wave-uniform, every lane's three arguments are >= 32 (the short series).
// (Round 6 also built the test-count terms from per-test histograms shared by the K prefixes of the cohort reference sets -- one reciprocal per distinct
// count instead of a digamma + trigamma evaluation per cell, review r5 item 2 -- and measured nothing: 13.95 against 13.95 ms for the stage, same box,
// alternating.  After the two changes above the stage is no longer bound by this kernel (4.6 of its 14 ms); the form is not in the tree.)
__device__ __forceinline__ void accumulate_cell_run(Acc& acc, double& pa, double& pb, double a, double b, double th, int y, int n, bool big)
{
  if (n <= 0) return;
  double x1, r1, q1, x2, r2, q2, x3, r3, q3;
  if (big) {
    x1 = a + (double)y; x2 = b + (double)(n - y); x3 = th + (double)n;
    digamma_trigamma_nolog_big(x1, r1, q1);
    digamma_trigamma_nolog_big(x2, r2, q2);
    digamma_trigamma_nolog_big(x3, r3, q3);
  } else {
    digamma_trigamma_nolog(a + (double)y, x1, r1, q1);
    digamma_trigamma_nolog(b + (double)(n - y), x2, r2, q2);
    digamma_trigamma_nolog(th + (double)n, x3, r3, q3);
  }
  const double i3 = frcp(x3);
  pa *= x1 * i3;
  pb *= x2 * i3;
  acc.ga += r1 - r3;
  acc.gb += r2 - r3;
  acc.haa += q1 - q3;
  acc.hab -= q3;
  acc.hbb += q2 - q3;
}

}  // namespace edfit
