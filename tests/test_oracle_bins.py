"""CPU checks of the phi.bins > 1 checker (oracle/bins_oracle.py, edo_fit_mle_groups): R's quantile(type 7),
seq(by=), the depth levels, approxfun -- against values worked out by hand from the R definitions -- and the
grouped MLE against the single-group one."""
import numpy as np
import pytest

from oracle import bins_oracle as bo
from oracle import edoracle as eo


def test_quantile_type7_and_seq():
    x = [10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110]          # n = 11: index = 1 + 10 p
    assert bo.r_quantile7(x, 0.85) == (1 - 0.5) * 90 + 0.5 * 100   # index 9.5
    assert bo.r_quantile7(x, 1.0) == 110
    assert bo.r_quantile7(x, 0.5) == 60
    assert bo.r_quantile7([5, 5, 5, 9], 0.85) == (1 - 0.55) * 5 + 0.55 * 9 or abs(bo.r_quantile7([5, 5, 5, 9], 0.85) - 7.2) < 1e-12
    s = bo.r_seq_by(0.0, 95.0, 95.0 / 3)
    assert s.size == 4 and s[0] == 0 and s[-1] == 95.0 and np.all(np.diff(s) > 0)
    assert np.array_equal(bo.r_seq_by(0.0, 0.0, 0.0), [0.0])


def test_depth_levels_and_interpolation():
    ref = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100], dtype=float)
    complete, quant = bo.depth_bins(ref, 3)                     # q85 = 85, edges 0, 42.5, 85, 101
    assert np.allclose(complete, [0, 42.5, 85, 101])
    assert list(quant) == [1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3]
    with pytest.raises(ValueError, match="Binning did not happen properly"):
        bo.depth_bins(np.full(20, 7.0), 3)
    mid = (complete[:3] + complete[1:]) / 2                     # 21.25, 63.75, 93
    y = np.array([0.01, 0.02, 0.04])
    v = bo.approx_linear(np.array([0.0, 21.25, 42.5, 63.75, 80.0, 93.0, 100.0]), mid, y)
    assert v[0] == 0.01 and v[1] == 0.01 and v[3] == 0.02 and v[5] == 0.04 and v[6] == 0.04
    assert abs(v[2] - 0.015) < 1e-15
    assert abs(v[4] - (0.02 + 0.02 * (80.0 - 63.75) / (93.0 - 63.75))) < 1e-15


def test_grouped_mle_reduces_to_single_group():
    rng = np.random.default_rng(3)
    n = 3000
    tot = rng.poisson(400, n)
    pp = rng.beta(0.12 * 150, 0.88 * 150, n)
    y = rng.binomial(tot, pp).astype(np.int32)
    r = (tot - y).astype(np.int32)
    phi1, p1, ll1, _ = eo.fit_mle(y, r)
    phig, pg, llg, _ = eo.fit_mle_groups(y, r, np.zeros(n, np.int32), 1)
    assert abs(phig[0] - phi1) < 1e-12 * phi1 and abs(pg - p1) < 1e-12
    # two arbitrary groups of the same population: both estimates near the pooled one, likelihood not lower
    grp = (np.arange(n) % 2).astype(np.int32)
    phi2, p2, ll2, _ = eo.fit_mle_groups(y, r, grp, 2)
    assert ll2 >= ll1 - 1e-9 * abs(ll1)
    assert np.all(np.abs(phi2 - phi1) / phi1 < 0.2)


def test_covariate_mle_reduces_to_intercept_only_and_recovers_a_slope():
    rng = np.random.default_rng(9)
    n = 4000
    gc = rng.uniform(-0.2, 0.2, n)
    tot = rng.poisson(600, n)
    p = 1 / (1 + np.exp(-(-2.0 + 1.5 * gc)))
    phi = 0.006
    y = rng.binomial(tot, rng.beta(p * (1 - phi) / phi, (1 - p) * (1 - phi) / phi)).astype(np.int32)
    r = (tot - y).astype(np.int32)
    b0, phi0, ll0, _ = eo.fit_mle_cov(y, r, np.zeros((n, 0)))
    phi1, p1, ll1, _ = eo.fit_mle(y, r)
    assert abs(phi0 - phi1) < 1e-12 * phi1 and abs(1 / (1 + np.exp(-b0[0])) - p1) < 1e-12
    b, ph, ll, _ = eo.fit_mle_cov(y, r, gc[:, None])
    assert ll > ll0 and abs(b[1] - 1.5) < 0.2 and abs(b[0] + 2.0) < 0.05 and abs(ph - phi) < 0.002
