#!/usr/bin/env python3
"""Generate the polynomial coefficients used by exomedepth_amd/csrc/ed_pmath.h.

The portable log/exp/sin are *our own* definitions (not glibc's, not ocml's): a fixed
sequence of IEEE-754 binary64 +,-,*,/,fma operations, so that gcc on the host and hipcc on
gfx950 produce bit-identical results.  This script derives near-minimax coefficients with
mpmath (Chebyshev interpolation at 60 digits) and prints them as C hex-float literals,
together with the measured approximation error, so the header can be regenerated/audited.

Usage: python tools/gen_pmath_coeffs.py            (prints a C fragment to stdout)
"""
import mpmath as mp

mp.mp.dps = 60


def hexf(x):
    return float(x).hex()


def fit(func, a, b, n):
    """Near-minimax degree-(n-1) polynomial coefficients (ascending) of func on [a,b]."""
    c = mp.chebyfit(func, [a, b], n)  # descending powers
    return list(reversed(c))


def max_rel_err(func, poly, a, b, npts=4001):
    worst = mp.mpf(0)
    for i in range(npts):
        x = a + (b - a) * mp.mpf(i) / (npts - 1)
        f = func(x)
        p = sum(c * x ** k for k, c in enumerate(poly))
        if f != 0:
            worst = max(worst, abs((p - f) / f))
    return worst



def log_table():
    """The table of the table-driven log (ed_plog_core_t): 128 sub-intervals of [45/64, 90/64), boundaries on a
    2^-8 grid below 1 and on a 2^-7 grid above (so that 1 is a boundary); row = invc = double(1/centre),
    logc = -log(invc) split into a multiple of 2^-42 and a remainder; LN2 split the same way, so that
    k*LN2_HI + logc_hi is exact in binary64."""
    OFF = mp.mpf(45) / 64       # 0.703125
    rows = []
    for i in range(128):
        if i < 76:
            c = OFF + (mp.mpf(i) + mp.mpf(1) / 2) * mp.mpf(2) ** -8
        else:
            c = 1 + (mp.mpf(i - 76) + mp.mpf(1) / 2) * mp.mpf(2) ** -7
        invc = mp.mpf(float(1 / c))
        logc = -mp.log(invc)
        hi = mp.nint(logc * mp.mpf(2) ** 42) / mp.mpf(2) ** 42
        lo = float(logc - hi)
        rows.append((float(invc), float(hi), lo))
    ln2 = mp.log(2)
    ln2hi = mp.nint(ln2 * mp.mpf(2) ** 42) / mp.mpf(2) ** 42
    ln2lo = float(ln2 - ln2hi)
    print("#define ED_PM_LOGT_LN2_HI %s" % float(ln2hi).hex())
    print("#define ED_PM_LOGT_LN2_LO %s" % float(ln2lo).hex())
    print("#define ED_PM_LOGT_N 128")
    print("#define ED_PM_LOGT_ROWS { \\")
    for k, (a, b, c) in enumerate(rows):
        print("  { %s, %s, %s }%s \\" % (a.hex(), b.hex(), c.hex(), "," if k < 127 else ""))
    print("}")
    # check exactness claims
    import math
    for a, b, c in rows:
        assert (b * 2 ** 42) == int(b * 2 ** 42)
    assert float(ln2hi) * 2 ** 42 == int(float(ln2hi) * 2 ** 42)


def main():
    out = []
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == 'logtable':
        log_table()
        return
    # ---- log: G(z) = (log((1+s)/(1-s)) - 2s) / (s*z),  z = s^2, |s| <= sqrt(2)-1 over sqrt(2)+1
    smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
    zmax = smax ** 2 * mp.mpf("1.0001")

    def G(z):
        if z == 0:
            return mp.mpf(2) / 3
        s = mp.sqrt(z)
        return (mp.log((1 + s) / (1 - s)) - 2 * s) / (s * z)

    for nlog in range(5, 12):
        cl = fit(G, mp.mpf(0), zmax, nlog)
        err = max_rel_err(G, [mp.mpf(float(c)) for c in cl], mp.mpf(0), zmax)
        if err < mp.mpf("1.2e-16"):  # floor set by rounding the coefficients to binary64
            break
    out.append("/* log: G(z)=(2atanh(s)-2s)/(s z), z=s^2 in [0,%s]; %d coeffs; max rel err of G %s"
               " (G contributes <1.5%% of the result) */" % (mp.nstr(zmax, 8), nlog, mp.nstr(err, 3)))
    out.append("#define ED_PM_LOG_NC %d" % nlog)
    out.append("#define ED_PM_LOG_COEFFS { " + ", ".join(hexf(c) for c in cl) + " }")

    # ---- exp: Q(r) = (exp(r)-1-r)/r^2 on |r| <= ln2/2 (+slack)
    rmax = mp.log(2) / 2 * mp.mpf("1.0002")

    def Q(r):
        if r == 0:
            return mp.mpf(1) / 2
        return (mp.exp(r) - 1 - r) / (r * r)

    for nexp in range(8, 16):
        ce = fit(Q, -rmax, rmax, nexp)
        err = max_rel_err(Q, [mp.mpf(float(c)) for c in ce], -rmax, rmax)
        # Q*r^2 <= 0.07 of the result: need err*0.07 < 2^-60
        if err * mp.mpf("0.07") < mp.mpf(2) ** -60:
            break
    out.append("/* exp: Q(r)=(e^r-1-r)/r^2 on |r|<=%s; %d coeffs; max rel err of Q %s */"
               % (mp.nstr(rmax, 8), nexp, mp.nstr(err, 3)))
    out.append("#define ED_PM_EXP_NC %d" % nexp)
    out.append("#define ED_PM_EXP_COEFFS { " + ", ".join(hexf(c) for c in ce) + " }")

    # ---- sin: P(w) = (sin(t)-t)/t^3, w=t^2, t in [0, pi/2 (+slack)]
    tmax = mp.pi / 2 * mp.mpf("1.001")
    wmax = tmax ** 2

    def P(w):
        if w == 0:
            return -mp.mpf(1) / 6
        t = mp.sqrt(w)
        return (mp.sin(t) - t) / (t * w)

    for nsin in range(8, 18):
        cs = fit(P, mp.mpf(0), wmax, nsin)
        err = max_rel_err(P, [mp.mpf(float(c)) for c in cs], mp.mpf(0), wmax)
        if err < mp.mpf("1.2e-16"):  # floor set by rounding the coefficients to binary64
            break
    out.append("/* sin: P(w)=(sin t - t)/t^3, w=t^2, t in [0,%s]; %d coeffs; max rel err of P %s */"
               % (mp.nstr(tmax, 8), nsin, mp.nstr(err, 3)))
    out.append("#define ED_PM_SIN_NC %d" % nsin)
    out.append("#define ED_PM_SIN_COEFFS { " + ", ".join(hexf(c) for c in cs) + " }")

    # ---- constants
    ln2 = mp.log(2)
    # ln2 split: hi has 32 significant bits so k*hi is exact for |k| < 2^21
    hi = mp.floor(ln2 * 2 ** 32) / 2 ** 32
    lo = ln2 - hi
    out.append("#define ED_PM_LN2_HI %s" % hexf(hi))
    out.append("#define ED_PM_LN2_LO %s" % hexf(lo))
    out.append("#define ED_PM_INV_LN2 %s" % hexf(1 / ln2))
    out.append("#define ED_PM_SQRT2 %s" % hexf(mp.sqrt(2)))
    print("\n".join(out))


if __name__ == "__main__":
    main()
