/* ed_dtab.h -- log-gamma DIFFERENCE tables for the table-driven emission mode (emit mode 1, edtab.inc).
 *
 * The emission of a cell is  log B(a1 + obs, a2 + tot - obs) - log B(a1, a2)  (reference src/CNV_estimate.cpp:44-50), and
 * log B(x, y) = lgamma(x) + lgamma(y) - lgamma(x + y) (reference src/beta.c:101-108), so with ref = tot - obs
 *     emission = D(a1, obs) + D(a2, ref) - D(a1 + a2, tot),      D(x, k) = lgamma(x + k) - lgamma(x) = sum_{i<k} log(x + i)
 * For a sample and a state (a1, a2) are constants and the counts are small integers: three tables per (sample, state),
 * each a prefix sum of logarithms.  This header defines how an entry is computed -- the same text is compiled by hipcc
 * into k_tab_build and by gcc into the checker's `edo_dtab` (tests/test_dtab.py compares it with mpmath), so "what a
 * table holds" is one definition:
 *     D[0] = 0,  D[k] = round_to_binary64( sum_{i<k} ddlog(fl(x0 + i)) ),
 * the sum carried in double-double, ddlog a logarithm with ~2^-74 absolute error.  An entry is therefore the correctly
 * rounded value of the exact sum of exact logarithms of the ROUNDED arguments fl(x0 + i) except when that sum lies within
 * ~2^-64 (relative) of a rounding boundary; measured against mpmath: <= 0.5001 ulp.
 *
 * Own algorithm: table-driven argument reduction on the 128-row table of ed_pmath.h (log x = k ln2 + log c + log1p(r),
 * r = z/c - 1 carried as a double-double), the quadratic term of log1p exact through fma, the tail a short polynomial.
 * Nothing here is taken from the reference (which has no such function) or from a libm.
 */
#ifndef ED_DTAB_H
#define ED_DTAB_H

#include "ed_pmath.h"

typedef struct { double hi, lo; } ed_dd;

ED_PM_FN ed_dd ed_dd_make(double hi, double lo) { ed_dd r; r.hi = hi; r.lo = lo; return r; }
/* error-free a + b */
ED_PM_FN ed_dd ed_two_sum(double a, double b)
{
  const double s = a + b, bb = s - a;
  return ed_dd_make(s, (a - (s - bb)) + (b - bb));
}
/* error-free a + b for |a| >= |b| (or a == 0) */
ED_PM_FN ed_dd ed_fast_two_sum(double a, double b)
{
  const double s = a + b;
  return ed_dd_make(s, b - (s - a));
}
/* double-double sum, relative error <= 2^-104 whatever the signs */
ED_PM_FN ed_dd ed_dd_add(ed_dd a, ed_dd b)
{
  ed_dd s = ed_two_sum(a.hi, b.hi);
  const ed_dd t = ed_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = ed_fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return ed_fast_two_sum(s.hi, s.lo);
}

/* double-double sum without the second renormalisation: absolute error <= 2^-104 (|a| + |b|) -- what a running sum of logarithms
 * needs (the entries are rounded to binary64 relative to their own size, and |D| >= 2^-40 max|partial sum| wherever it matters) */
ED_PM_FN ed_dd ed_dd_add_fast(ed_dd a, ed_dd b)
{
  ed_dd s = ed_two_sum(a.hi, b.hi);
  s.lo += a.lo + b.lo;
  return ed_fast_two_sum(s.hi, s.lo);
}
/* round(a + b) of two double-doubles to binary64 (to within 2^-104 of the sum before the rounding) */
ED_PM_FN double ed_dd_add_hi(ed_dd a, ed_dd b)
{
  const ed_dd s = ed_two_sum(a.hi, b.hi);
  return s.hi + (s.lo + (a.lo + b.lo));
}

/* log(x) as a double-double, x positive, normal, finite.  T: the 128 x 3 table ED_PM_LOGT_ROWS (row i at T[3 i]).
 *   x = 2^k z, z in [45/64, 90/64); row i = (invc, logc_hi, logc_lo), logc = -log(invc) to ~2^-97
 *   z invc = p + pe exactly (fma), q = p - 1 exactly, r = q + pe as (rh, rl)
 *   log x = [k LN2_HI + logc_hi]  (exact: both are multiples of 2^-42)  + rh - rh^2/2   (error-free sums)
 *           + { k LN2_LO + logc_lo + rl - (rh^2 error)/2 - rh rl + rh^3 (1/3 - rh/4 + ... - rh^7/10) }
 *   |r| < 2^-8: the first dropped term r^11/11 is below 2^-91; the bracket is accumulated in binary64 (|bracket| < 2^-22). */
ED_PM_FN ed_dd ed_ddlog_t(double x, const double* T)
{
  const uint64_t ix = ed_pm_bits(x);
  const uint64_t tmp = ix - 0x3fe6800000000000ULL;                    /* 45/64 */
  const int i = (int)((tmp >> 45) & 127);
  const double kd = (double)(int)((int64_t)tmp >> 52);
  const double z = ed_pm_from_bits(ix - (tmp & 0xfff0000000000000ULL));
  const double invc = T[3 * i], lch = T[3 * i + 1], lcl = T[3 * i + 2];
  const double p = z * invc;
  const double pe = ed_pm_fma(z, invc, -p);
  const double q = p - 1.0;                                           /* exact */
  const double rh = q + pe;
  const double rl = (q - rh) + pe;                                    /* |q| >= |pe| or q == 0 */
  const double w = ed_pm_fma(kd, ED_PM_LOGT_LN2_HI, lch);             /* exact */
  const ed_dd s1 = ed_two_sum(w, rh);
  const double p2 = rh * rh;
  const double e2 = ed_pm_fma(rh, rh, -p2);
  const ed_dd s2 = ed_two_sum(s1.hi, -0.5 * p2);
  double g = -1.0 / 10.0;
  g = ed_pm_fma(g, rh, 1.0 / 9.0);
  g = ed_pm_fma(g, rh, -1.0 / 8.0);
  g = ed_pm_fma(g, rh, 1.0 / 7.0);
  g = ed_pm_fma(g, rh, -1.0 / 6.0);
  g = ed_pm_fma(g, rh, 1.0 / 5.0);
  g = ed_pm_fma(g, rh, -1.0 / 4.0);
  g = ed_pm_fma(g, rh, 1.0 / 3.0);
  double lo = ed_pm_fma(kd, ED_PM_LOGT_LN2_LO, lcl);
  lo += rl;
  lo += ed_pm_fma(-rh, rl, -0.5 * e2);
  lo = ed_pm_fma(p2 * rh, g, lo);
  lo += s1.lo;
  lo += s2.lo;
  return ed_fast_two_sum(s2.hi, lo);
}

/* Arguments the tables accept: the logarithm above wants a positive normal number, and x0 + i must stay one */
ED_PM_FN int ed_dtab_shape_ok(double x0) { return x0 >= 0x1p-1000 && x0 <= 0x1p+1000; }

/* The definition, sequentially: out[k] = D(x0, k), k = 0 .. n-1 (the checker's edo_dtab; the device builds the same sums
 * with a parallel scan, i.e. in another association of double-double additions: equal to ~2^-100, hence equal after
 * rounding except on a measure-zero set). */
ED_PM_FN void ed_dtab_fill_seq(double x0, int64_t n, double* out, const double* T)
{
  ed_dd acc = ed_dd_make(0.0, 0.0);
  for (int64_t k = 0; k < n; ++k) {
    out[k] = acc.hi;
    acc = ed_dd_add_fast(acc, ed_ddlog_t(x0 + (double)k, T));
  }
}

/* emission = (D1 + D2) - D3 of three table entries: the exact sum of the three binary64 numbers, rounded once */
ED_PM_FN double ed_dtab_combine(double d1, double d2, double d3)
{
  const ed_dd s = ed_two_sum(d1, d2);
  const ed_dd t = ed_two_sum(s.hi, -d3);
  return t.hi + (t.lo + s.lo);
}

#endif /* ED_DTAB_H */
