/* ed_pmath.h -- portable, bit-reproducible log / exp / sin for binary64.
 *
 * Why this exists.  The hot path (beta-binomial emissions, reference src/CNV_estimate.cpp:44-85 ->
 * src/beta.c:49-114 -> src/VP_gamma.c) calls libm's log() and exp().  glibc's and ROCm's (ocml)
 * implementations differ in the last bit for a fraction of arguments, and the Viterbi forward pass
 * compares sums of emissions with a strict '>' (reference src/hmm.cpp:79-84), so a 1-ulp difference
 * can in principle flip a back-pointer.  To make "GPU result == CPU checker result" a bit-exact
 * statement rather than a statistical one, the three transcendental functions the path needs are
 * DEFINED here as a fixed sequence of IEEE-754 binary64 operations (+ - * / and fma, all correctly
 * rounded on x86-64 and on gfx950) plus integer bit manipulation.  The same text is compiled by gcc
 * (for the checker in oracle/) and by hipcc (for the kernels), and gives the same bits on both.
 *
 * These are our own algorithms (classic argument reduction + near-minimax polynomials whose
 * coefficients are produced by tools/gen_pmath_coeffs.py); nothing here is taken from glibc, ocml
 * or the reference.  Accuracy (checked in tests/test_pmath.py against mpmath): log < 0.8 ulp, exp < 0.9 ulp,
 * sin (cold path only) < 1.6 ulp.
 *
 * Build requirement: -ffp-contract=off on both compilers (every fused operation below is an explicit
 * ed_pm_fma); on the host also -mfma so that the builtin is a single vfmadd (glibc's software fma()
 * is used otherwise: same bits, slower).
 */
#ifndef ED_PMATH_H
#define ED_PMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define ED_PM_FN __host__ __device__ static __inline__ __attribute__((always_inline))
#else
#define ED_PM_FN static inline
#endif

/* ---- coefficients: output of tools/gen_pmath_coeffs.py (do not edit by hand) ---- */
/* log: G(z)=(2atanh(s)-2s)/(s z), z=s^2 in [0,0.029440195]; 8 coeffs; max rel err of G 5.55e-17 (G contributes <1.5% of the result) */
#define ED_PM_LOG_NC 8
#define ED_PM_LOG_COEFFS { 0x1.5555555555555p-1, 0x1.9999999999a38p-2, 0x1.2492492476c87p-2, 0x1.c71c720160b47p-3, 0x1.745cf901605f7p-3, 0x1.3b1c360763b8ap-3, 0x1.0fbe83c2cbcf3p-3, 0x1.0c04595972ab8p-3 }
/* exp: Q(r)=(e^r-1-r)/r^2 on |r|<=0.3466429; 11 coeffs; max rel err of Q 4.76e-18 */
#define ED_PM_EXP_NC 11
#define ED_PM_EXP_COEFFS { 0x1.0000000000000p-1, 0x1.5555555555557p-3, 0x1.5555555555556p-5, 0x1.11111111100d8p-7, 0x1.6c16c16c162d2p-10, 0x1.a01a01abe9ce8p-13, 0x1.a01a01a6d9931p-16, 0x1.71de022bd5558p-19, 0x1.27e4db6121beep-22, 0x1.af4df5750ec30p-26, 0x1.1f730a202ec17p-29 }
/* sin: P(w)=(sin t - t)/t^3, w=t^2, t in [0,1.5723671]; 8 coeffs; max rel err of P 6.69e-17 */
#define ED_PM_SIN_NC 8
#define ED_PM_SIN_COEFFS { -0x1.5555555555555p-3, 0x1.1111111111107p-7, -0x1.a01a01a018a6bp-13, 0x1.71de3a5453a9cp-19, -0x1.ae6455a012592p-26, 0x1.612400c8c470dp-33, -0x1.ae5109fea6b7ap-41, 0x1.899eb3575b7f2p-49 }
#define ED_PM_LN2_HI 0x1.62e42fee00000p-1
#define ED_PM_LN2_LO 0x1.a39ef35793c76p-33
#define ED_PM_INV_LN2 0x1.71547652b82fep+0
#define ED_PM_SQRT2 0x1.6a09e667f3bcdp+0
/* ---- end generated ---- */

ED_PM_FN uint64_t ed_pm_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
ED_PM_FN double ed_pm_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
ED_PM_FN double ed_pm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* fma(a, b, k) with a compile-time constant addend k (Horner steps).  Same operation, same bits; on the device the
 * constant is handed over in scalar registers: left to itself the compiler picks v_fmac_f64, whose addend must
 * already sit in the destination VGPRs, and spends two extra vector moves per coefficient to put it there. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ static __inline__ __attribute__((always_inline)) double ed_pm_fma_k(double a, double b, double k)
{
  double d;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
}
#else
ED_PM_FN double ed_pm_fma_k(double a, double b, double k) { return __builtin_fma(a, b, k); }
#endif
ED_PM_FN double ed_pm_inf(void) { return ed_pm_from_bits(0x7ff0000000000000ULL); }
ED_PM_FN double ed_pm_nan(void) { return ed_pm_from_bits(0x7ff8000000000000ULL); }

/* natural logarithm.  x = 2^k * m, m in (sqrt2/2, sqrt2]; f = m-1; s = f/(2+f);
 * log(m) = f - f^2/2 + s*(f^2/2 + R(s^2)),  R(z) = z*G(z) ~ 2atanh(s)/s - 2. */
ED_PM_FN double ed_plog(double x)
{
  const double c[ED_PM_LOG_NC] = ED_PM_LOG_COEFFS;
  uint64_t u = ed_pm_bits(x);
  int k = 0;
  if (!(x > 0.0)) {               /* zero, negative, NaN */
    if (x == 0.0) return -ed_pm_inf();
    return ed_pm_nan();
  }
  if (u >= 0x7ff0000000000000ULL) return x;   /* +inf */
  if (u < 0x0010000000000000ULL) {            /* subnormal: renormalise */
    x = x * 0x1p54;
    u = ed_pm_bits(x);
    k = -54;
  }
  k += (int)(u >> 52) - 1023;
  double m = ed_pm_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > ED_PM_SQRT2) { m = m * 0.5; k += 1; }
  const double f = m - 1.0;                    /* exact */
  const double s = f / (2.0 + f);
  const double z = s * s;
  double g = c[ED_PM_LOG_NC - 1];
  for (int i = ED_PM_LOG_NC - 2; i >= 0; --i) g = ed_pm_fma_k(g, z, c[i]);
  const double R = z * g;
  const double hfsq = (0.5 * f) * f;
  const double dk = (double)k;
  const double w = ed_pm_fma(s, hfsq + R, dk * ED_PM_LN2_LO);
  const double v = f - (hfsq - w);
  return ed_pm_fma(dk, ED_PM_LN2_HI, v);       /* dk*LN2_HI is exact (32-bit constant) */
}

/* exponential.  x = k ln2 + r, |r| <= ln2/2; e^r = 1 + (r + r^2 Q(r)); result scaled by 2^k in two
 * exact-power-of-two multiplications so that subnormal results round once. */
ED_PM_FN double ed_pexp(double x)
{
  const double c[ED_PM_EXP_NC] = ED_PM_EXP_COEFFS;
  if (x != x) return x;
  if (x > -0x1p-5 && x < 0x1p-5) {
    /* short series (the one case the hot path has: exp of a Stirling tail, |x| < 0.0084): e^x = 1 + (x + x^2 Q),
     * Q = 1/2 + x/6 + ... + x^5/5040; the first dropped term, x^8/40320, is below 2.3e-17 */
    double q = 1.0 / 5040.0;
    q = ed_pm_fma_k(q, x, 1.0 / 720.0);
    q = ed_pm_fma_k(q, x, 1.0 / 120.0);
    q = ed_pm_fma_k(q, x, 1.0 / 24.0);
    q = ed_pm_fma_k(q, x, 1.0 / 6.0);
    q = ed_pm_fma_k(q, x, 0.5);
    return 1.0 + ed_pm_fma(x * x, q, x);
  }
  if (x > 0x1.62e42fefa39efp+9) return ed_pm_inf();     /* > log(DBL_MAX) */
  if (x < -0x1.74910d52d3051p+9) return 0.0;            /* < log(2^-1075) */
  const double t = x * ED_PM_INV_LN2;
  const double kd = (t + 0x1.8p52) - 0x1.8p52;          /* round to nearest integer, ties to even */
  double r = ed_pm_fma(-kd, ED_PM_LN2_HI, x);           /* exact */
  r = ed_pm_fma(-kd, ED_PM_LN2_LO, r);
  double q = c[ED_PM_EXP_NC - 1];
  for (int i = ED_PM_EXP_NC - 2; i >= 0; --i) q = ed_pm_fma_k(q, r, c[i]);
  const double p = ed_pm_fma(r * r, q, r);
  double e = 1.0 + p;
  const int k = (int)kd;
  const int k1 = k >> 1;            /* arithmetic shift: floor(k/2) */
  const int k2 = k - k1;
  e = e * ed_pm_from_bits((uint64_t)(1023 + k1) << 52);
  e = e * ed_pm_from_bits((uint64_t)(1023 + k2) << 52);
  return e;
}

/* sine on [0, pi] only -- the one place the path needs it is the reflection branch of log-gamma
 * for 0.02 <= x < 0.5 (reference src/VP_gamma.c:1180-1183 evaluates sin(pi*(1-x)), :1247-1249
 * sin(pi*x)).  For t > pi/2 the argument is folded with a two-part pi: u = (PI_HI - t) + PI_LO, the
 * first subtraction being exact.  Outside [0, pi] the result is NaN. */
#define ED_PM_PI_HI 0x1.921fb54442d18p+1
#define ED_PM_PI_LO 0x1.1a62633145c07p-53
ED_PM_FN double ed_psin_0pi(double t)
{
  const double c[ED_PM_SIN_NC] = ED_PM_SIN_COEFFS;
  if (!(t >= 0.0 && t <= ED_PM_PI_HI)) return ed_pm_nan();
  if (t > 0.5 * ED_PM_PI_HI) t = (ED_PM_PI_HI - t) + ED_PM_PI_LO;
  const double w = t * t;
  double p = c[ED_PM_SIN_NC - 1];
  for (int i = ED_PM_SIN_NC - 2; i >= 0; --i) p = ed_pm_fma_k(p, w, c[i]);
  return ed_pm_fma(t * w, p, t);
}

#endif /* ED_PMATH_H */
