"""The table-driven emission mode's entry definition (exomedepth_amd/csrc/ed_dtab.h), compiled by gcc into the checker, against
mpmath: the double-double logarithm, the prefix sums D(x0, k) = sum_{i<k} log(fl(x0 + i)) and the three-term combination."""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def test_ddlog_absolute_accuracy(oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(-600, 600, 600)), rng.uniform(0.5, 2, 600), 1 + rng.uniform(-2e-2, 2e-2, 600),
                        rng.uniform(1, 1e5, 600), [2.0 ** -1000, 2.0 ** 1000, 1.0, 45 / 64, 90 / 64]])
    hi, lo = oracle.ddlog(x)
    worst = max(float(abs(mp.mpf(float(h)) + mp.mpf(float(l)) - mp.log(mp.mpf(float(xi))))) for xi, h, l in zip(x, hi, lo))
    assert worst < 2.0 ** -72, worst
    assert np.all(np.abs(lo) <= np.spacing(np.abs(hi)))          # normalised pairs


def test_table_entries_are_correctly_rounded(oracle):
    rng = np.random.default_rng(4)
    for x0, n in ((22.3, 12000), (0.37, 3000), (178.123, 20000), (1.0000001, 2000), (1e6 + 0.5, 6000), (1e-3, 1500)):
        d = oracle.dtab(x0, n)
        assert d[0] == 0.0
        want = set(int(k) for k in np.concatenate([np.arange(1, 30), rng.integers(1, n, 150), [n - 1]]))
        acc, worst = mp.mpf(0), 0.0
        for k in range(n):
            if k in want and acc != 0:
                ulp = mp.mpf(2) ** (mp.floor(mp.log(abs(acc), 2)) - 52)
                worst = max(worst, float(abs(mp.mpf(float(d[k])) - acc) / ulp))
            acc += mp.log(mp.mpf(float(np.float64(x0) + np.float64(k))))
        assert worst <= 0.5001, (x0, worst)


def test_against_lgamma_differences(oracle):
    """D(x, k) = lgamma(x + k) - lgamma(x): the identity the mode rests on (reference src/beta.c:101-108), to the rounding of x + i"""
    for x0 in (40.8, 160.2, 201.0, 0.05):
        d = oracle.dtab(x0, 5000)
        for k in (1, 7, 100, 999, 4999):
            t = mp.loggamma(mp.mpf(x0) + k) - mp.loggamma(mp.mpf(x0))
            assert abs(mp.mpf(float(d[k])) - t) <= 4e-16 * abs(t) + 1e-18, (x0, k)


def test_combine_is_the_rounded_exact_sum(oracle):
    rng = np.random.default_rng(5)
    d1 = rng.uniform(0, 600, 4000); d2 = rng.uniform(0, 6000, 4000); d3 = d1 + d2 + rng.uniform(-400, 0, 4000)
    got = oracle.dtab_combine(d1, d2, d3)
    for a, b, c, g in zip(d1, d2, d3, got):
        t = mp.mpf(float(a)) + mp.mpf(float(b)) - mp.mpf(float(c))
        assert abs(mp.mpf(float(g)) - t) <= abs(t) * mp.mpf(2) ** -52


def test_stirling_tails_against_mpmath(oracle):
    """D(x0, k) beyond the tables (ed_dtab_tail): lgamma(fl(x0 + k)) - lgamma(x0) to ~1 ulp, for shape parameters from 1e-3 to 1e5 and counts from the
    first served one (64) to 2^24; lgamma(x0) itself (ed_dtab_lg0) to double-double accuracy"""
    rng = np.random.default_rng(6)
    worst = 0.0
    for x0 in (1.5e-3, 0.37, 1.0, 2.0, 22.3, 40.8, 178.123, 201.0, 1999.5, 7.0e4 + 0.25):
        hi, lo = oracle.dtab_lg0(x0)
        t0 = mp.loggamma(mp.mpf(x0))
        assert abs(mp.mpf(hi) + mp.mpf(lo) - t0) <= mp.mpf(2) ** -60 * max(abs(t0), 1), x0      # (its series part is summed in binary64: ~1e-19 absolute, five orders below an entry's ulp)
        k = np.unique(np.concatenate([[64, 65, 100, 1000, 2476, 2 ** 24], np.exp(rng.uniform(np.log(64), np.log(2.0 ** 24), 300)).astype(np.int64)])).astype(np.float64)
        got = oracle.dtab_tail(x0, k)
        for kk, g in zip(k, got):
            z = mp.mpf(float(np.float64(x0) + np.float64(kk)))            # the rounded argument, as the reference forms it
            t = mp.loggamma(z) - t0
            # the series forms lgamma(z) and subtracts: what it can keep is an ulp of the LARGER of |lgamma(z)| and |D| (a table's entry: half an ulp of D
            # itself -- for counts far below the shape parameter the tables are the better of the two, and the conditioning rule of the tail samples,
            # edtab.inc: tab_cond_bound with tails, counts (x0 + k) log(x0 + k) for it)
            big = max(abs(t), abs(mp.loggamma(z)))
            ulp = mp.mpf(2) ** (mp.floor(mp.log(big, 2)) - 52)
            worst = max(worst, float(abs(mp.mpf(float(g)) - t) / ulp))
    assert worst < 2.5, worst


def test_tails_continue_the_tables(oracle):
    """where a table ends and the series takes over the two agree to the last bits (both approximate the same function: 0.5 ulp and ~1 ulp)"""
    for x0 in (0.9, 22.3, 178.123, 201.0):
        d = oracle.dtab(x0, 9000)
        k = np.arange(64, 9000, dtype=np.float64)
        t = oracle.dtab_tail(x0, k)
        assert np.max(np.abs(t - d[64:]) / np.spacing(np.abs(d[64:]))) <= 3.0
