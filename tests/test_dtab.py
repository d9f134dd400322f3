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
