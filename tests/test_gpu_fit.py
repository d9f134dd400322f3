"""Dispersion fit (kernel group K5) against the checker's high-precision MLE.

Parity status: UNPINNED against the reference -- phi/expected come from aod::betabin there
(R/class_definition.R:118), a third-party package outside the reference tree.  What is tested is that
the device reaches the maximum of the documented likelihood: (phi, p) within FIT_REL_TOL of the
checker's long-double Newton solution, and within Nelder-Mead's own tolerance of the aod stand-in.
"""
import os

import numpy as np
import pytest

from test_gpu_parity import eval_sf

pytestmark = pytest.mark.gpu

FIT_REL_TOL = 1e-8


def test_device_digamma_trigamma(edlib, oracle):
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-2), np.log(1e7), 200_000)), [0.5, 1.0, 9.999, 10.0, 10.001]])
    psi, psi1 = oracle.psi(x)
    got0 = eval_sf(edlib, 6, x)
    got1 = eval_sf(edlib, 7, x)
    # psi crosses zero at x0 = 1.4616...: absolute bound there, relative elsewhere
    assert np.max(np.abs(got0 - psi) / np.maximum(np.abs(psi), 1.0)) < 5e-15
    assert np.max(np.abs(got1 - psi1) / np.abs(psi1)) < 5e-15


def test_short_series_of_the_batched_fit(edlib):
    """the reference-set searches' batched per-cell fit (k_fit_accum) evaluates digamma / trigamma of arguments >= 32 with four Bernoulli terms instead of
    seven: the same values to the last bits, also through the cell routine with lanes of either kind side by side"""
    rng = np.random.default_rng(8)
    x = np.exp(rng.uniform(np.log(32.0), np.log(1e7), 5000))
    assert np.max(np.abs(eval_sf(edlib, 15, x) - eval_sf(edlib, 6, x)) / np.abs(eval_sf(edlib, 6, x))) < 5e-16
    assert np.max(np.abs(eval_sf(edlib, 16, x) - eval_sf(edlib, 7, x)) / eval_sf(edlib, 7, x)) < 5e-16
    a = np.exp(rng.uniform(np.log(0.5), np.log(3000), 8192)); y = rng.integers(0, 4000, 8192).astype(float)
    d = eval_sf(edlib, 17, a, y) - eval_sf(edlib, 18, a, y)
    assert np.max(np.abs(d)) < 1e-14


def _fit_case(edlib, oracle, E, S, seed, geometry=None, **kw):
    from exomedepth_amd import synth
    chrom_off, start, end = synth.exon_design(E, 4, seed)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=5, **kw)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    if geometry is not None:
        batch.set_fit_histograms(geometry)
    dphi = edlib.DeviceArray(np.zeros(S))
    dexp = edlib.DeviceArray(np.zeros(S))
    batch.fit(test, ref, dphi, dexp)
    from exomedepth_amd._lib import check, lib
    check(lib().ed_synchronize(None))
    gphi, gexp = dphi.to_host(), dexp.to_host()
    batch.close(); plan.close()
    for s in range(S):
        ophi, op, _, _ = oracle.fit_mle(test[:, s], ref[:, s])
        assert abs(gphi[s] - ophi) / ophi < FIT_REL_TOL, (s, gphi[s], ophi)
        assert abs(gexp[s] - op) / op < FIT_REL_TOL, (s, gexp[s], op)
    return test, ref, gphi, gexp


def test_fit_matches_high_precision_mle(edlib, oracle):
    _fit_case(edlib, oracle, E=6000, S=70, seed=41)


def test_fit_short_and_low_depth(edlib, oracle):
    _fit_case(edlib, oracle, E=300, S=5, seed=42)                      # fewer exons than one coarse stride pass
    _fit_case(edlib, oracle, E=4000, S=9, seed=43, mean_depth=8.0)     # low depth: small shape arguments


def test_fit_vs_neldermead_standin(edlib, oracle):
    test, ref, gphi, gexp = _fit_case(edlib, oracle, E=5000, S=3, seed=44)
    for s in range(3):
        nphi, npp, _ = oracle.fit_nm(test[:, s], ref[:, s])
        assert abs(gphi[s] - nphi) / nphi < 2e-2    # optim()'s reltol 1.5e-8 on the objective ~ 1e-3..1e-2 on phi
        assert abs(gexp[s] - npp) / npp < 2e-3


def test_fit_then_run_end_to_end(edlib, oracle):
    """fit -> emissions -> Viterbi on the device; the checker, fed the device's (phi, p), must agree bit for bit."""
    from exomedepth_amd import synth
    E, S = 3000, 6
    chrom_off, start, end = synth.exon_design(E, 3, 51)
    test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, 51, n_segments=3, mean_depth=70.0)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dphi = edlib.DeviceArray(np.zeros(S)); dexp = edlib.DeviceArray(np.zeros(S))
    batch.fit(test, ref, dphi, dexp)
    batch.run(test, ref, dphi, dexp)
    ll, path = batch.loglik(), batch.path()
    gphi, gexp = dphi.to_host(), dexp.to_host()
    batch.close(); plan.close()
    for s in range(S):
        ell, _ = oracle.get_loglike_matrix(gphi[s], gexp[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.PORTABLE)
        assert np.array_equal(ll[:, :, s].view(np.int64), np.ascontiguousarray(ell).view(np.int64))
        ep, _ = oracle.callcnvs(ell, chrom_off, start, end)
        assert np.array_equal(path[:, s].astype(np.int8), ep)


def test_fit_subset_for_speed(edlib, oracle):
    """Scalar subset.for.speed (R/class_definition.R:107-113): the fit sees rows seq(1, nrow, by = floor(nrow / n))
    only.  Batched entry (by-strided view) and the per-sample mirror against the checker's MLE on those rows."""
    from exomedepth_amd import synth
    from exomedepth_amd._lib import check, lib
    E, S, n_sub = 9001, 5, 700
    by = E // n_sub
    chrom_off, start, end = synth.exon_design(E, 3, 61)
    test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, 61, n_segments=3)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dphi = edlib.DeviceArray(np.zeros(S)); dexp = edlib.DeviceArray(np.zeros(S))
    batch.fit(test, ref, dphi, dexp, by=by)
    check(lib().ed_synchronize(None))
    gphi, gexp = dphi.to_host(), dexp.to_host()
    batch.close(); plan.close()
    rows = np.arange(0, E, by)
    assert rows.size == (E - 1) // by + 1
    for s in range(S):
        ophi, op, _, _ = oracle.fit_mle(test[rows, s], ref[rows, s])
        assert abs(gphi[s] - ophi) / ophi < FIT_REL_TOL
        assert abs(gexp[s] - op) / op < FIT_REL_TOL
    x = edlib.ExomeDepth(test[:, 0].astype(float), ref[:, 0].astype(float), subset_for_speed=n_sub)
    assert abs(x.phi[0] - gphi[0]) / gphi[0] < 1e-12 and x.phi.size == E
    assert abs(x.expected[0] - gexp[0]) / gexp[0] < 1e-12
    # vector form: explicit 1-based rows, non-existing ones dropped
    idx = np.concatenate([rows + 1, [0, E + 5]])
    y = edlib.ExomeDepth(test[:, 0].astype(float), ref[:, 0].astype(float), subset_for_speed=idx)
    assert y.phi[0] == x.phi[0]


@pytest.mark.parametrize("hist", [True, False])
def test_fit_slow_start_case(edlib, oracle, hist):
    """A case the randomised sweep (tools/fuzz_parity.py) found: 266 exons at depth ~5, MLE phi = 3.1e-4, moment start
    clamped to 1e-4, where the likelihood is not concave in lambda.  A Hessian-scaled gradient step crawled there
    (1.4 % per iteration) and both iteration budgets ran out; the fallback step is now Newton-sized and capped."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "fit_slow_start_case.npz"))
    t, r = d["test"].astype(np.int32), d["ref"].astype(np.int32)
    E, S = t.size, 5
    T = np.repeat(t[:, None], S, axis=1).copy(); R = np.repeat(r[:, None], S, axis=1).copy()
    plan = edlib.Plan(np.array([0, E], dtype=np.int32), np.arange(E, dtype=np.int32) * 100, np.arange(E, dtype=np.int32) * 100 + 50)
    batch = edlib.Batch(plan, S)
    batch.set_fit_histograms(hist)
    dphi = edlib.DeviceArray(np.zeros(S)); dexp = edlib.DeviceArray(np.zeros(S))
    batch.fit(T, R, dphi, dexp)
    from exomedepth_amd._lib import check, lib
    check(lib().ed_synchronize(None))
    gphi, gexp = dphi.to_host(), dexp.to_host()
    batch.close(); plan.close()
    ophi, op, _, _ = oracle.fit_mle(t, r)
    # a + b = 3 200 here: the digamma differences of the gradient cancel ~4 digits more than at phi ~ 5e-3, and the
    # binary64 digamma (5e-15 relative) shows: 1e-8 relative on phi observed, 1e-7 asserted
    assert np.all(np.abs(gphi - ophi) < 1e-7 * ophi) and np.all(np.abs(gexp - op) < FIT_REL_TOL * op)


@pytest.mark.parametrize("geometry", [8, 4, 2])
@pytest.mark.parametrize("depth", [60.0, 250.0, 900.0])
def test_fit_every_histogram_geometry(edlib, oracle, geometry, depth):
    """The three histogram geometries (8 / 4 / 2 samples per workgroup of k_fit_hist: unit bins to 4096 / 8192 / 16384)
    are picked from the data's depth; here each is forced on shallow, medium and deep data, so that every one of them
    meets counts inside its bins, in the second-level bins, in the lists, and (deep data on the small geometry) lists
    that run out.  S = 13: not a multiple of any workgroup's sample count."""
    _fit_case(edlib, oracle, E=7000, S=13, seed=int(depth) + geometry, geometry=geometry, mean_depth=depth)


def test_fit_geometry_follows_depth(edlib, oracle):
    """Automatic choice: same answers (checked against the MLE) whichever geometry the depth selects."""
    for depth in (100.0, 260.0, 500.0):
        _fit_case(edlib, oracle, E=5000, S=9, seed=900 + int(depth), mean_depth=depth)


def test_fit_geometry_hint_is_only_a_hint(edlib, oracle):
    """A batch launches the histogram geometry its PREVIOUS fit's depth points to; which of the launched kernels runs is
    decided on the device from the current data.  Shallow, then deep, then deep, then shallow data through one batch:
    the second and fourth fits run on a geometry chosen for other data -- same answers."""
    from exomedepth_amd import synth
    from exomedepth_amd._lib import check, lib
    E, S = 6000, 7
    chrom_off, start, end = synth.exon_design(E, 3, 77)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dphi = edlib.DeviceArray(np.zeros(S)); dexp = edlib.DeviceArray(np.zeros(S))
    for k, depth in enumerate((60.0, 700.0, 700.0, 60.0)):
        test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, 770 + k, n_segments=4, mean_depth=depth)
        batch.fit(test, ref, dphi, dexp)
        check(lib().ed_synchronize(None))
        gphi, gexp = dphi.to_host(), dexp.to_host()
        for s in range(S):
            ophi, op, _, _ = oracle.fit_mle(test[:, s], ref[:, s])
            assert abs(gphi[s] - ophi) / ophi < FIT_REL_TOL and abs(gexp[s] - op) / op < FIT_REL_TOL, (k, depth, s, gphi[s], ophi)
    batch.close(); plan.close()


def test_fit_counts_beyond_the_histogram_range(edlib, oracle):
    """Deep data: every reference count lies beyond the 4096 bins, the overflow regions run out, and the Newton
    kernel sums those samples cell by cell (same launch).  A moderately deep case uses bins and overflow lists together."""
    _fit_case(edlib, oracle, E=3000, S=6, seed=45, mean_depth=3000.0)
    _fit_case(edlib, oracle, E=5000, S=5, seed=46, mean_depth=400.0)


def _fit_both_layouts(edlib, test, ref, by=1):
    """(phi, expected, unconverged) from [exons][samples] counts and from sample-major counts"""
    E, S = test.shape
    plan = edlib.Plan(np.array([0, E], np.int32), np.arange(E, dtype=np.int32) * 100, np.arange(E, dtype=np.int32) * 100 + 50)
    out = []
    for layout in (0, 1):
        b = edlib.Batch(plan, S)
        if layout:
            b.set_emit_mode(2); b.set_counts_layout(1)
        t, r = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
        dphi, dexp = edlib.DeviceArray(np.zeros(S)), edlib.DeviceArray(np.zeros(S))
        b.fit(edlib.DeviceArray(t), edlib.DeviceArray(r), dphi, dexp, by=by)
        out.append((dphi.to_host(), dexp.to_host(), b.fit_unconverged()[0]))
        b.close()
    plan.close()
    return out


def test_per_cell_fit_on_ill_conditioned_columns(edlib, oracle):
    """The per-cell Newton passes (histograms off: the form the reference-set searches run batched) stop when a step over all exons falls below
    1e-6 (2e-5, kFitStepTol, in the reference-set searches' batched form only -- round 5 had applied it here too), leaving ~C 1e-12 against the
    fit's 1e-8: pinned here on the columns where C is largest -- tiny dispersions, few exons, low depth, a proportion near 0 or 1 -- against the
    checker's long-double maximum-likelihood fit."""
    rng = np.random.default_rng(2026)
    cases = []
    for E in (40, 150, 600, 9000):                               # (9000: the coarse passes run too)
        S = 48
        phi = np.exp(rng.uniform(np.log(2e-5), np.log(0.3), S))
        p = np.concatenate([np.exp(rng.uniform(np.log(2e-3), np.log(0.5), S // 2)), 1 - np.exp(rng.uniform(np.log(2e-3), np.log(0.5), S - S // 2))])
        depth = np.exp(rng.uniform(np.log(15.0), np.log(3000.0), S))
        n = rng.poisson(depth[None, :] * rng.lognormal(0, 0.6, (E, 1))).astype(np.int64)
        a, b = p * (1 - phi) / phi, (1 - p) * (1 - phi) / phi
        q = rng.beta(a[None, :], b[None, :], (E, S))
        test = rng.binomial(n, q).astype(np.int32)
        cases.append((test, (n - test).astype(np.int32)))
    worst = 0.0
    for test, ref in cases:
        E, S = test.shape
        plan = edlib.Plan(np.array([0, E], dtype=np.int32), np.arange(E, dtype=np.int32) * 100, np.arange(E, dtype=np.int32) * 100 + 50)
        b = edlib.Batch(plan, S)
        b.set_fit_histograms(0)
        dphi, dexp = edlib.DeviceArray(np.zeros(S)), edlib.DeviceArray(np.zeros(S))
        b.fit(test, ref, dphi, dexp)
        nu, _ = b.fit_unconverged()
        gphi, gexp = dphi.to_host(), dexp.to_host()
        b.close(); plan.close()
        checked = beyond = 0
        for s in range(S):
            ophi, op, _, _ = oracle.fit_mle(test[:, s], ref[:, s])
            if not (2e-6 < ophi < 0.6):                          # (the fit's own bounds on phi: other tests cover the pinned columns)
                continue
            # (binary64 gradient sums locate a dispersion below ~1.5e-3 to 2e-14 / phi^2 only -- DESIGN.md 4.5; the checker sums in long double)
            err = max(abs(gphi[s] - ophi) / ophi, abs(gexp[s] - op) / op) / max(1.0, 2e-6 / (ophi * ophi))
            checked += 1
            if err < 1e-8:
                worst = max(worst, err)
            else:
                beyond += 1
                print("beyond 1e-8: E %d column %d phi %.6g (checker %.6g) expected %.6g (checker %.6g) err %.3g" % (E, s, gphi[s], ophi, gexp[s], op, err))
        # a column the fit DECLARES converged is within 1e-8; the ones it reports as not converged within its ten passes (40-exon columns whose
        # likelihood is nearly flat in the dispersion) are the only ones that may lie beyond
        assert checked >= S // 2 and nu <= S // 6 and beyond <= nu, (E, checked, nu, beyond)
    assert worst < 1e-8, worst


def test_column_pinned_at_the_dispersion_floor_still_converges_its_mean(edlib, oracle):
    """nearly binomial columns (the likelihood's maximum lies below the phi >= 1e-6 the fit allows): phi is pinned at the bound and
    the expected proportion is the maximum GIVEN that phi -- in both layouts, to the fit's tolerance of the checker's"""
    rng = np.random.default_rng(12)
    E, S = 300, 24
    n = rng.integers(150, 600, size=(E, S))
    p = rng.uniform(0.05, 0.2, S)
    test = rng.binomial(n, p[None, :]).astype(np.int32)          # no over-dispersion at all
    ref = (n - test).astype(np.int32)
    got = _fit_both_layouts(edlib, test, ref)
    for phi, exp, unconv in got:
        assert unconv <= 2            # (a maximum just above the bound -- phi ~ 1.4e-6 here -- sits where the psi-gradient is rounding noise: its steps never fall below the tolerance)
        pinned = phi <= 1.0000002e-6
        assert pinned.sum() >= S // 3
        for s in np.flatnonzero(pinned):
            # at phi = 1e-6 the model is the binomial to ~1e-4 relative in the likelihood's curvature: its maximum in p is sum(y) / sum(n) to ~1e-6
            want = test[:, s].sum() / n[:, s].sum()
            assert abs(exp[s] - want) / want < 2e-5, (s, exp[s], want)
    both = (got[0][0] <= 1.0000002e-6) & (got[1][0] <= 1.0000002e-6)
    assert np.max(np.abs(got[0][1][both] - got[1][1][both]) / got[0][1][both]) < 1e-8


def test_sample_major_histograms_with_bins_that_leave_the_lds_early(edlib, oracle):
    """columns of 120 000 exons on three count values: a bin passes 16 384 cells inside a chunk and is carried out of the 16-bit LDS bins
    (k_fit_hist_sm); second-level ranges and lists in play (reference counts beyond 4 096); the estimate equals the [E][S] form's and the
    checker's"""
    rng = np.random.default_rng(13)
    E, S = 120_000, 5
    test = (rng.integers(0, 3, size=(E, S)) * 11 + 20).astype(np.int32)
    ref = (rng.integers(0, 3, size=(E, S)) * 2500 + 300).astype(np.int32)       # 300 / 2 800 / 5 300: first level, first level, second level
    ref[::97, 2] = 20_000                                                     # ... and a few beyond both levels: the list
    (p0, e0, u0), (p1, e1, u1) = _fit_both_layouts(edlib, test, ref)
    assert u0 == 0 and u1 == 0
    assert np.max(np.abs(p1 - p0) / p0) < 1e-7 and np.max(np.abs(e1 - e0) / e0) < 1e-9
    for s in (0, 2):
        op, oe = oracle.fit_mle_hist(test[:, s], ref[:, s])[:2]
        assert abs(p1[s] - op) / op < 1e-6 and abs(e1[s] - oe) / oe < 1e-8
