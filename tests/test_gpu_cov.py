"""Covariates in the mean model (`data` + `formula = cbind(test, reference) ~ x1 + ...`, reference
R/class_definition.R:86-118, :168) through the C-ABI.

Parity status: the coefficients come from aod::betabin in the reference (not in the reference tree): UNPINNED, compared
at tolerance with the checker's long-double MLE of the documented likelihood (oracle edo_fit_mle_cov).  The likelihood,
Viterbi path and call table are compared bit for bit with the checker fed the same per-exon `expected`.
"""
import numpy as np
import pytest

from test_gpu_parity import bits

pytestmark = pytest.mark.gpu


def _cohort(E, S, K, seed, depth=120.0):
    from exomedepth_amd import synth
    rng = np.random.default_rng(seed)
    chrom_off, start, end = synth.exon_design(E, 3, seed)
    X = np.stack([rng.uniform(0.3, 0.7, E) - 0.5, rng.normal(0.0, 1.0, E), rng.uniform(-1, 1, E)], axis=1)[:, :K]
    lam = rng.lognormal(np.log(depth), 0.6, E)
    test = np.zeros((E, S), dtype=np.int32); ref = np.zeros((E, S), dtype=np.int32)
    truth = []
    for s in range(S):
        beta = np.concatenate([[rng.uniform(-2.4, -1.6)], rng.uniform(-0.8, 0.8, K) * np.array([2.0, 0.15, 0.3])[:K]])
        phi = rng.uniform(0.003, 0.012)
        p = 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))
        tot = rng.poisson(lam * 9)
        pp = rng.beta(p * (1 - phi) / phi, (1 - p) * (1 - phi) / phi)
        y = rng.binomial(tot, pp)
        test[:, s] = y; ref[:, s] = tot - y
        truth.append((beta, phi))
    return chrom_off, start, end, X, test, ref, truth


@pytest.mark.parametrize("K", [0, 1, 3])
def test_fit_cov_matches_checker(edlib, oracle, K):
    E, S = 6000, 5
    chrom_off, start, end, X, test, ref, truth = _cohort(E, S, K, 300 + K)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dbeta = edlib.DeviceArray(np.zeros((K + 1, S))); dphi = edlib.DeviceArray(np.zeros(S))
    batch.fit_cov(test, ref, X, dbeta, dphi)
    beta, phi = dbeta.to_host(), dphi.to_host()
    batch.close(); plan.close()
    for s in range(S):
        obeta, ophi, _, _ = oracle.fit_mle_cov(test[:, s], ref[:, s], X)
        assert np.all(np.abs(beta[:, s] - obeta) < 1e-7 * np.maximum(1.0, np.abs(obeta))), (s, beta[:, s], obeta)
        assert abs(phi[s] - ophi) < 1e-6 * ophi, (s, phi[s], ophi)
        if K == 0:                                  # intercept only: the plain model
            ophi1, op1, _, _ = oracle.fit_mle(test[:, s], ref[:, s])
            assert abs(phi[s] - ophi1) < 1e-6 * ophi1 and abs(1 / (1 + np.exp(-beta[0, s])) - op1) < 1e-8
        # the fit recovers the planted coefficients to statistical accuracy
        assert np.all(np.abs(beta[:, s] - truth[s][0]) < 0.2)


def test_run_cov_pipeline_parity(edlib, oracle):
    E, S, K = 4000, 4, 2
    chrom_off, start, end, X, test, ref, _ = _cohort(E, S, K, 311, depth=60.0)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dbeta = edlib.DeviceArray(np.zeros((K + 1, S))); dphi = edlib.DeviceArray(np.zeros(S))
    batch.fit_cov(test, ref, X, dbeta, dphi)
    batch.run_cov(test, ref, X, dbeta, dphi)
    ll, path, calls, info = batch.loglik(), batch.path(), batch.calls(), batch.call_info()
    expd = batch.expected_cov(X, dbeta)
    phi = dphi.to_host()
    batch.set_fused(True)
    with pytest.raises(edlib.EdError):
        batch.run_cov(test, ref, X, dbeta, dphi)
    batch.close(); plan.close()
    assert len(info) == len(calls)
    for s in range(S):
        ell, _ = oracle.get_loglike_matrix(np.full(E, phi[s]), expd[:, s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.PORTABLE)
        assert np.array_equal(bits(ll[:, :, s]), bits(ell)), "sample %d" % s
        epath, ecalls = oracle.callcnvs(ell, chrom_off, start, end)
        assert np.array_equal(path[:, s].astype(np.int8), epath)
        m = calls[calls["sample"] == s]
        assert len(m) == len(ecalls) and np.array_equal(m["start_exon"] + 1, ecalls[:, 0].astype(np.int64))
    # reads.expected of the decoration uses the per-exon expected (R/class_definition.R:398)
    for c, f in list(zip(calls, info))[:50]:
        s0, a, b = int(c["sample"]), int(c["start_exon"]), int(c["end_exon"])
        assert f["reads_expected"] == int(np.sum((test[a:b + 1, s0] + ref[a:b + 1, s0]) * expd[a:b + 1, s0]))


def test_mirror_formula_with_covariates(edlib, oracle):
    E, K = 3000, 1
    chrom_off, start, end, X, test, ref, _ = _cohort(E, 1, K, 320)
    t, r = test[:, 0].astype(float), ref[:, 0].astype(float)
    x = edlib.ExomeDepth(t, r, data={"GC": X[:, 0]}, formula="cbind(test, reference) ~ GC")
    obeta, ophi, _, _ = oracle.fit_mle_cov(test[:, 0], ref[:, 0], X)
    oexp = 1 / (1 + np.exp(-(obeta[0] + X[:, 0] * obeta[1])))
    assert np.all(np.abs(x.expected - oexp) < 1e-7) and abs(x.phi[0] - ophi) < 1e-6 * ophi
    chrom = np.concatenate([[str(c + 1)] * int(chrom_off[c + 1] - chrom_off[c]) for c in range(3)])
    x.CallCNVs(chrom, start, end, np.array(["e%d" % i for i in range(E)]))
    ell, _ = oracle.get_loglike_matrix(x.phi, x.expected, test[:, 0] + ref[:, 0], test[:, 0], 1.0, oracle.PORTABLE)
    assert np.array_equal(bits(x.likelihood), bits(ell))
    epath, _ = oracle.callcnvs(ell, chrom_off, start, end)
    assert np.array_equal(x.Viterbi_path.astype(np.int8), epath)
    with pytest.raises(ValueError):
        edlib.ExomeDepth(t, r, formula="cbind(test, reference) ~ GC")          # no data
    with pytest.raises(NotImplementedError):
        edlib.ExomeDepth(t, r, data={"GC": X[:, 0]}, formula="cbind(test, reference) ~ GC * len")


@pytest.mark.parametrize("E,S,seed,depth", [(953, 1, 119733737, 200.0), (1713, 3, 761797754, 200.0), (3857, 30, 856114134, 200.0)])
def test_fit_cov_hard_starts(edlib, oracle, E, S, seed, depth):
    """Cases the randomised sweep (tools/fuzz_more.py) broke an earlier iteration on: a strong slope on a narrow covariate
    (the intercept-only optimum is then a SADDLE of the full model: the plain Newton direction is not an ascent direction),
    and a first dispersion step that overshoots (coordinate-wise clipping turned the direction).  The third one sent the
    checker's own line search to phi = 1e-23 before it was boxed like the device."""
    K = 3
    from exomedepth_amd import synth
    chrom_off, start, end = synth.exon_design(E, 1, seed)
    r2 = np.random.default_rng(seed)
    X = np.ascontiguousarray(np.stack([r2.uniform(-0.2, 0.2, E), r2.normal(0, 1, E), r2.uniform(-1, 1, E)], axis=1))
    lam = r2.lognormal(np.log(depth), 0.6, E)
    test = np.zeros((E, S), dtype=np.int32); ref = np.zeros((E, S), dtype=np.int32)
    for s in range(S):
        beta = np.concatenate([[r2.uniform(-2.4, -1.6)], r2.uniform(-0.8, 0.8, K) * np.array([2.0, 0.15, 0.3])])
        phi_t = r2.uniform(0.003, 0.012)
        pe = 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))
        tot = r2.poisson(lam * 9)
        yy = r2.binomial(tot, r2.beta(pe * (1 - phi_t) / phi_t, (1 - pe) * (1 - phi_t) / phi_t))
        test[:, s] = yy; ref[:, s] = tot - yy
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dbeta = edlib.DeviceArray(np.zeros((K + 1, S))); dphi = edlib.DeviceArray(np.zeros(S))
    batch.fit_cov(test, ref, X, dbeta, dphi)
    bt, ph = dbeta.to_host(), dphi.to_host()
    batch.close(); plan.close()
    for s in range(S):
        obeta, ophi, _, _ = oracle.fit_mle_cov(test[:, s], ref[:, s], X)
        assert 1e-4 < ophi < 0.1, (s, ophi)
        assert np.all(np.abs(bt[:, s] - obeta) < 1e-6 * np.maximum(1.0, np.abs(obeta))), (s, bt[:, s], obeta)
        assert abs(ph[s] - ophi) < 1e-6 * ophi, (s, ph[s], ophi)
