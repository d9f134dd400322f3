"""Randomised sweep of the secondary modes against the CPU checkers: fused kernel (bit-identical to the default
path), covariates in the mean model, phi.bins > 1 (edges / interpolation / likelihood bits; per-level dispersions at tolerance), and
select.reference.set (order, bins, medians, choice exact; statistics at tolerance).
    python tools/fuzz_more.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo
from oracle import bins_oracle as bo
from oracle import refset_oracle as ro

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
t0 = time.time()
n = {"fused": 0, "bins": 0, "bins_rejected": 0, "refset": 0, "cov": 0}
while time.time() - t0 < budget:
    mode = rng.choice(["fused", "bins", "refset", "cov"])
    seed = int(rng.integers(1 << 30))
    if mode == "fused":
        S = int(rng.choice([1, 5, 16, 17, 64, 100])); C = int(rng.integers(1, 5)); E = int(rng.integers(C, 900))
        chrom_off, start, end = synth.exon_design(E, C, seed)
        test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=float(rng.choice([20.0, 300.0])))
        plan = ed.Plan(chrom_off, start, end)
        out = []
        for fused in (False, True):
            batch = ed.Batch(plan, S)
            batch.set_fused(fused)
            batch.run(test, ref, phi, p)
            out.append((batch.loglik(), batch.path(), batch.calls()))
            batch.close()
        plan.close()
        assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and np.array_equal(out[0][1], out[1][1]), ("fused", E, S, C, seed)
        assert out[0][2].tobytes() == out[1][2].tobytes(), ("fused calls", E, S, C, seed)
    elif mode == "bins":
        S = int(rng.choice([1, 4, 9, 70])); C = int(rng.integers(1, 4)); E = int(rng.integers(300, 3000)); B = int(rng.integers(2, 9))
        chrom_off, start, end = synth.exon_design(E, C, seed)
        test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=float(rng.choice([30.0, 200.0])))
        plan = ed.Plan(chrom_off, start, end); batch = ed.Batch(plan, S)
        dphib = ed.DeviceArray(np.zeros((B, S))); dedges = ed.DeviceArray(np.zeros((B + 1, S))); dexp = ed.DeviceArray(np.zeros(S))
        try:
            batch.fit_bins(test, ref, B, dphib, dedges, dexp)
        except ed.EdError as e:
            assert "Binning did not happen properly" in str(e)
            ok = [True] * S
            for s in range(S):
                try:
                    bo.depth_bins(ref[:, s], B)
                except ValueError:
                    ok[s] = False
            assert not all(ok), ("bins: device rejected, checker accepts", E, S, B, seed)
            n["bins_rejected"] += 1
            batch.close(); plan.close()
            continue
        batch.run_bins(test, ref, B, dphib, dedges, dexp)
        ll, path = batch.loglik(), batch.path()
        philin = batch.phi_linear(ref, B, dphib, dedges)
        phib, edges, ex = dphib.to_host(), dedges.to_host(), dexp.to_host()
        batch.close(); plan.close()
        for s in rng.choice(S, size=min(S, 3), replace=False):
            ophi, op, olin, ocomp = bo.fit_bins(test[:, s], ref[:, s], B)
            assert np.array_equal(bits(edges[:, s]), bits(ocomp)), ("bins edges", E, S, B, seed, s)
            if np.all(ophi > 1e-4) and np.all(ophi < 0.4):
                assert np.max(np.abs(phib[:, s] - ophi) / ophi) < 1e-5, ("bins phi", E, S, B, seed, s, phib[:, s], ophi)
            ell, _ = eo.get_loglike_matrix(philin[:, s], np.full(E, ex[s]), test[:, s] + ref[:, s], test[:, s], 1.0, eo.PORTABLE)
            assert np.array_equal(bits(ll[:, :, s]), bits(ell)), ("bins loglik", E, S, B, seed, s)
            epath, _ = eo.callcnvs(ell, chrom_off, start, end)
            assert np.array_equal(path[:, s].astype(np.int8), epath), ("bins path", E, S, B, seed, s)
    elif mode == "cov":
        S = int(rng.choice([1, 3, 66])); C = int(rng.integers(1, 4)); E = int(rng.integers(800, 4000)); K = int(rng.integers(0, 4))
        chrom_off, start, end = synth.exon_design(E, C, seed)
        r2 = np.random.default_rng(seed)
        X = np.stack([r2.uniform(-0.2, 0.2, E), r2.normal(0, 1, E), r2.uniform(-1, 1, E)], axis=1)[:, :K]
        lam = r2.lognormal(np.log(float(rng.choice([40.0, 200.0]))), 0.6, E)
        test = np.zeros((E, S), dtype=np.int32); ref = np.zeros((E, S), dtype=np.int32)
        for s in range(S):
            beta = np.concatenate([[r2.uniform(-2.4, -1.6)], r2.uniform(-0.8, 0.8, K) * np.array([2.0, 0.15, 0.3])[:K]])
            phi_t = r2.uniform(0.003, 0.012)
            pe = 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))
            tot = r2.poisson(lam * 9)
            yy = r2.binomial(tot, r2.beta(pe * (1 - phi_t) / phi_t, (1 - pe) * (1 - phi_t) / phi_t))
            test[:, s] = yy; ref[:, s] = tot - yy
        plan = ed.Plan(chrom_off, start, end); batch = ed.Batch(plan, S)
        dbeta = ed.DeviceArray(np.zeros((K + 1, S))); dphi = ed.DeviceArray(np.zeros(S))
        batch.fit_cov(test, ref, X, dbeta, dphi)
        batch.run_cov(test, ref, X, dbeta, dphi)
        ll, path = batch.loglik(), batch.path()
        expd = batch.expected_cov(X, dbeta)
        bt, ph = dbeta.to_host(), dphi.to_host()
        batch.close(); plan.close()
        for s in rng.choice(S, size=min(S, 2), replace=False):
            obeta, ophi, _, _ = eo.fit_mle_cov(test[:, s], ref[:, s], X)
            assert np.all(np.abs(bt[:, s] - obeta) < 1e-6 * np.maximum(1.0, np.abs(obeta))), ("cov beta", E, S, K, seed, s, bt[:, s], obeta)
            assert abs(ph[s] - ophi) < max(1e-6, 1e-13 / ophi ** 2) * ophi, ("cov phi", E, S, K, seed, s, ph[s], ophi)
            ell, _ = eo.get_loglike_matrix(np.full(E, ph[s]), expd[:, s], test[:, s] + ref[:, s], test[:, s], 1.0, eo.PORTABLE)
            assert np.array_equal(bits(ll[:, :, s]), bits(ell)), ("cov loglik", E, S, K, seed, s)
            epath, _ = eo.callcnvs(ell, chrom_off, start, end)
            assert np.array_equal(path[:, s].astype(np.int8), epath), ("cov path", E, S, K, seed, s)
    else:
        E = int(rng.integers(2000, 9000)); R = int(rng.integers(2, 14))
        lam = rng.lognormal(np.log(60.0), 0.7, E)
        test = rng.poisson(lam).astype(np.int32)
        sig = np.linspace(0.02, 0.4, R)[rng.permutation(R)]
        refs = np.stack([rng.poisson(lam * rng.lognormal(0.0, s_, E) * rng.uniform(0.7, 1.3)) for s_ in sig], axis=1).astype(np.int32)
        bl = rng.integers(60, 600, E).astype(np.float64) if rng.random() < 0.5 else None
        red = int(rng.choice([0, 0, 1500]))
        got = ed.select_reference_set(test, refs, bin_length=bl, n_bins_reduced=red)
        try:
            exp = ro.select_reference_set(test, refs, bin_length=bl, n_bins_reduced=red)
        except ZeroDivisionError:   # the checker's fit ran a binomial-looking prefix down to phi == 0 exactly
            exp = {"phi": np.zeros(1)}
        st = got["summary.stats"]
        if np.nanmin(exp["phi"]) < 1e-8:
            # the checker ran a binomial-looking prefix down to phi ~ 1e-16, where ITS power sum (log-Betas of arguments
            # ~1e15) is numerical noise (expected.BF 1e13 seen); the device stops at phi = 1e-6.  Not a comparable case.
            n["refset_skipped"] = n.get("refset_skipped", 0) + 1
            continue
        if len(got["reference.choice"]) != exp["n_chosen"]:
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez_compressed("gpurun_out/fuzz_refset_case.npz", test=test, refs=refs, bl=(bl if bl is not None else np.zeros(0)), red=red,
                                stats=st, exp_bf=exp["expected_BF"], exp_phi=exp["phi"], exp_p=exp["mean_p"], exp_med=exp["median_depth"])
        assert got["n.bins"] == exp["n_bins"] and np.array_equal(st["ref_index"], exp["order"]), ("refset order", E, R, seed)
        assert len(got["reference.choice"]) == exp["n_chosen"], ("refset choice", E, R, seed, len(got["reference.choice"]), exp["n_chosen"])
        for mine, theirs, tol in (("phi", "phi", 1e-6), ("mean_p", "mean_p", 1e-7), ("median_depth", "median_depth", 0.0),
                                  ("expected_BF", "expected_BF", 1e-6)):
            a, b = st[mine], exp[theirs]
            assert np.array_equal(np.isnan(a), np.isnan(b)), ("refset nan", mine, E, R, seed)
            m = ~np.isnan(b)
            if mine != "median_depth":           # a binomial-looking prefix: the device stops at its floor 1e-6, the checker runs on to ~0
                m &= ~(np.nan_to_num(exp["phi"], nan=1.0) < 1e-5)
            tol_i = np.maximum(tol, 1e-13 / np.nan_to_num(exp["phi"], nan=1.0) ** 2)   # ~2e-14 / phi^2
            assert np.all(np.abs(a[m] - b[m]) <= (tol_i * np.abs(b))[m]), ("refset", mine, E, R, seed, a, b)
    n[mode] += 1
print("fuzz_more ok:", n, "%.0f s" % (time.time() - t0))
