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

/* ---- beyond the tables: D(x0, k) for large k from Stirling's series (round 6) ---------------------------------------------------------------
 * A table costs an entry per count; deep sequencing (reference counts in the thousands and tens of thousands) outgrows what a workgroup can hold in
 * LDS and then what is worth building at all.  For k >= 64 the same function is cheap to evaluate directly:
 *     D(x0, k) = lgamma(z) - lgamma(x0),   z = fl(x0 + k)   (the reference's own rounded argument, src/CNV_estimate.cpp:49),
 *     lgamma(z) = (z - 1/2) log z - z + log(2 pi)/2 + 1/(12 z) - 1/(360 z^3) + 1/(1260 z^5) - ...      (the next term is below 2e-16 for z >= 64)
 * with lgamma(x0) a per-(sample, state, table) constant carried as a double-double (ed_dtab_lg0).  What is kept of the precision: the product
 * (z - 1/2) log z exactly (fma), the sums as TwoSums, the logarithm to its own 0.52 ulp -- z times that is the error that remains, ~1 ulp of D
 * (the table entries: 0.5 ulp).  An emission still is ed_dtab_combine of three such numbers.
 * The same text is compiled into the emission kernel and into the checker (edo_dtab_tail); tests/test_dtab.py compares both with mpmath. */
#define ED_DTAB_HALF_LOG_2PI_HI 0x1.d67f1c864beb5p-1
#define ED_DTAB_HALF_LOG_2PI_LO -0x1.65b5a1b7ff5dfp-55
#define ED_DTAB_TAIL_FROM 64            /* tails serve indices k >= this (every table window is at least this long) */

/* lgamma(x0) as a double-double, x0 positive and normal: shifted up by 64 -- lgamma(x0) = lgamma(x0 + 64) - sum_{i<64} log(x0 + i), the sum in
 * double-double with ed_ddlog_t, lgamma(x0 + 64) from the series above with seven terms and a double-double logarithm.  Sequential: one definition
 * for the device (a lane per state in k_tab_build) and the checker. */
ED_PM_FN ed_dd ed_dtab_lg0(double x0, const double* T)
{
  /* (the arguments x0 + i are NOT exact here, unlike a table's, whose definition is the sum over the rounded arguments: what the sum misses of
   *  log(x0 + i) is (x0 + i - fl(x0 + i)) / fl(x0 + i) to first order, the next order being below 2^-106) */
  ed_dd acc = ed_dd_make(0.0, 0.0);
  double fix = 0.0;
  for (int i = 0; i < 64; ++i) {
    const ed_dd a = ed_two_sum(x0, (double)i);
    acc = ed_dd_add_fast(acc, ed_ddlog_t(a.hi, T));
    fix += a.lo / a.hi;
  }
  acc = ed_dd_add(acc, ed_dd_make(fix, 0.0));
  const ed_dd zs = ed_two_sum(x0, 64.0);
  const double z = zs.hi;                                  /* lgamma(x0 + 64) = lgamma(z) + zs.lo psi(z) */
  const ed_dd lz = ed_ddlog_t(z, T);
  const double zm = z - 0.5;                               /* exact: z >= 64 */
  /* (z - 1/2) (lz.hi + lz.lo): the leading product exactly, the rest in binary64 */
  const double p = zm * lz.hi;
  const double pe = ed_pm_fma(zm, lz.hi, -p);
  ed_dd g = ed_two_sum(p, -z);
  g.lo += pe + zm * lz.lo;
  const double w = 1.0 / z, w2 = w * w;
  double ser = 1.0 / 156.0;
  ser = ed_pm_fma(ser, w2, -691.0 / 360360.0);
  ser = ed_pm_fma(ser, w2, 1.0 / 1188.0);
  ser = ed_pm_fma(ser, w2, -1.0 / 1680.0);
  ser = ed_pm_fma(ser, w2, 1.0 / 1260.0);
  ser = ed_pm_fma(ser, w2, -1.0 / 360.0);
  ser = ed_pm_fma(ser, w2, 1.0 / 12.0);
  g.lo += ser * w + ED_DTAB_HALF_LOG_2PI_LO;
  g.lo += zs.lo * (lz.hi - 0.5 * w);
  g = ed_dd_add(ed_fast_two_sum(g.hi, g.lo), ed_dd_make(ED_DTAB_HALF_LOG_2PI_HI, 0.0));
  return ed_dd_add(g, ed_dd_make(-acc.hi, -acc.lo));
}

/* D(x0, k), k >= ED_DTAB_TAIL_FROM, given lg0 = ed_dtab_lg0(x0) */
ED_PM_FN double ed_dtab_tail(double x0, double lg0_hi, double lg0_lo, double k, const double* T)
{
  const double z = x0 + k;
  const double lz = ed_plog_core_t(z, 0, T);
  const double zm = z - 0.5;                               /* exact */
  const double p = zm * lz;
  const double pe = ed_pm_fma(zm, lz, -p);
  const ed_dd a = ed_two_sum(p, -z);                       /* (z - 1/2) log z - z */
  const ed_dd b = ed_two_sum(a.hi, -lg0_hi);               /* ... - lgamma(x0) */
  const double w = 1.0 / z, w2 = w * w;
  double ser = 1.0 / 1260.0;
  ser = ed_pm_fma(ser, w2, -1.0 / 360.0);
  ser = ed_pm_fma(ser, w2, 1.0 / 12.0);
  const double small = ((ED_DTAB_HALF_LOG_2PI_HI + ser * w) + (ED_DTAB_HALF_LOG_2PI_LO - lg0_lo)) + ((pe + a.lo) + b.lo);
  return b.hi + small;
}

#endif /* ED_DTAB_H */
