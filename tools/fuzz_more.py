"""Randomised sweep of the secondary modes against the CPU checkers: fused kernel (bit-identical to the default
path), phi.bins > 1 (edges / interpolation / likelihood bits; per-level dispersions at tolerance), and
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
n = {"fused": 0, "bins": 0, "bins_rejected": 0, "refset": 0}
while time.time() - t0 < budget:
    mode = rng.choice(["fused", "bins", "refset"])
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
    else:
        E = int(rng.integers(2000, 9000)); R = int(rng.integers(2, 14))
        lam = rng.lognormal(np.log(60.0), 0.7, E)
        test = rng.poisson(lam).astype(np.int32)
        sig = np.linspace(0.02, 0.4, R)[rng.permutation(R)]
        refs = np.stack([rng.poisson(lam * rng.lognormal(0.0, s_, E) * rng.uniform(0.7, 1.3)) for s_ in sig], axis=1).astype(np.int32)
        bl = rng.integers(60, 600, E).astype(np.float64) if rng.random() < 0.5 else None
        red = int(rng.choice([0, 0, 1500]))
        got = ed.select_reference_set(test, refs, bin_length=bl, n_bins_reduced=red)
        exp = ro.select_reference_set(test, refs, bin_length=bl, n_bins_reduced=red)
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
