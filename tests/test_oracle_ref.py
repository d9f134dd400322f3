"""Container-only: the checker against the reference's own special functions on large random grids.
Skipped where oracle/_ref (built from /root/reference by oracle/Makefile) is absent -- the committed
golden vectors in tests/golden/ carry the same evidence to the GPU box."""
import numpy as np
import pytest

from oracle import edoracle as eo

pytestmark = pytest.mark.skipif(not eo.ref_available(), reason="oracle/_ref not built (no /root/reference)")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def test_lnbeta_bitwise_on_random_grid(oracle):
    rng = np.random.default_rng(99)
    n = 300_000
    x = np.exp(rng.uniform(np.log(1e-3), np.log(1e7), n))
    y = np.exp(rng.uniform(np.log(1e-3), np.log(1e7), n))
    assert np.array_equal(bits(oracle.lnbeta(x, y, oracle.LIBM)), bits(oracle.ref_call2("gsl_sf_lnbeta", x, y)))
    a = rng.uniform(0.05, 300, n) + rng.integers(0, 500, n)
    b = rng.uniform(0.5, 3000, n) + rng.integers(0, 3000, n)
    assert np.array_equal(bits(oracle.lnbeta(a, b, oracle.LIBM)), bits(oracle.ref_call2("gsl_sf_lnbeta", a, b)))


def test_unary_functions_bitwise(oracle):
    rng = np.random.default_rng(98)
    x = np.exp(rng.uniform(np.log(1e-6), np.log(1e12), 200_000))
    assert np.array_equal(bits(oracle.sf("lngamma_e", x, oracle.LIBM)[0]), bits(oracle.ref_call1("gsl_sf_lngamma", x)))
    assert np.array_equal(bits(oracle.sf("gammastar", x, oracle.LIBM)[0]), bits(oracle.ref_call1("gsl_sf_gammastar", x)))
    z = np.concatenate([rng.uniform(-0.99, 5, 100_000), rng.uniform(-1e-2, 1e-2, 100_000)])
    assert np.array_equal(bits(oracle.sf("log_1plusx", z, oracle.LIBM)[0]), bits(oracle.ref_call1("gsl_sf_log_1plusx", z)))


def test_checker_psi_vs_reference_psi(oracle):
    x = np.exp(np.random.default_rng(97).uniform(np.log(1e-2), np.log(1e7), 100_000))
    psi, psi1 = oracle.psi(x)
    r0 = oracle.ref_call1("gsl_sf_psi", x)
    r1 = oracle.ref_call1("gsl_sf_psi_1", x)
    assert np.max(np.abs(psi - r0) / np.maximum(np.abs(r0), 1e-3)) < 1e-13
    assert np.max(np.abs(psi1 - r1) / np.abs(r1)) < 1e-14


def test_lngamma_sgn_negative_arguments_bitwise(oracle):
    """gsl_sf_lngamma_sgn_e for x < 0 (src/VP_gamma.c:1244-1276): reflection, next to -1 and next to -N (lngamma_sgn_sing,
    :795-894, with gsl_sf_lnfact / psi_int / psi_1_int / psi_n -> hzeta behind it), far negative."""
    rng = np.random.default_rng(96)
    x = np.concatenate([-rng.uniform(0.02, 200, 200_000), -1 + rng.uniform(-0.0149, 0.0149, 20_000),
                        -rng.integers(2, 3000, 100_000) + rng.uniform(-0.0149, 0.0149, 100_000),
                        -rng.integers(170, 2_000_000, 20_000) + rng.uniform(-0.0149, 0.0149, 20_000),
                        -np.exp(rng.uniform(np.log(1e3), np.log(2e9), 20_000)), rng.uniform(-0.02, 0.02, 5_000)])   # (below INT_MIN + 2 the reference raises EROUND)
    x = x[x != np.floor(x)]
    rv, rs, rst = oracle.ref_lngamma_sgn(x)
    v, s, st = oracle.lngamma_sgn(x, oracle.LIBM)
    assert np.all(rst == 0) and np.array_equal(st, rst)
    assert np.array_equal(s, rs) and np.array_equal(bits(v), bits(rv))
    pv, ps, pst = oracle.lngamma_sgn(x, oracle.PORTABLE)
    big = np.abs(rv) > 1e-3
    assert np.array_equal(ps, rs) and np.max(np.abs(pv[big] - rv[big]) / np.abs(rv[big])) < 1e-10
    assert np.max(np.abs(pv[~big] - rv[~big]), initial=0.0) < 1e-13


def test_lnbeta_negative_arguments_with_positive_sign_bitwise(oracle):
    """gsl_sf_lnbeta on (x, y) with a negative non-integer argument and B(x, y) > 0 (no GSL error: the reference build can be
    called, its gsl_error being unresolved): libm flavour bit for bit; the sign rule (:43-45) through the sites."""
    rng = np.random.default_rng(95)
    n = 100_000
    x = -rng.uniform(0.02, 30, n)
    y = rng.uniform(-30, 60, n)
    ok = (x != np.floor(x)) & (y != np.floor(y)) & (y != 0) & ((x + y) != np.floor(x + y))
    x, y = x[ok], y[ok]
    v, sites = oracle.lnbeta_sites(x, y, oracle.LIBM)
    clean = sites == 0
    assert clean.sum() > 20_000 and (sites == 1 << 11).sum() > 20_000     # both signs of B occur
    assert np.array_equal(bits(v[clean]), bits(oracle.ref_call2("gsl_sf_lnbeta", x[clean], y[clean])))
    assert np.all(np.isnan(v[sites == 1 << 11]))
