"""Randomised sweep of round 3's additions against what they must equal:
  * the cohort pipeline (ed_cohort_run_host: random slab sizes, slabs in flight, layouts, wire formats, options) against the batch
    interface on the same data, bit for bit;
  * fit mode 1 (aod-nm on the device) against the checker's nmmin, within the search's tolerance;
  * ed_cohort_select_reference_sets against one ed_select_reference_set call per sample: identical choices;
  * the depth-binned model through the pipeline (option phi_bins: fit issued blind, settled at the first wait; declined slabs done
    again per cell; an empty level = the run's error) against ed_batch_fit_bins + ed_batch_run_bins.
    python tools/fuzz_cohort.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
from exomedepth_amd._lib import check, lib
from oracle import edoracle as eo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
t0 = time.time()
n = {"cohort": 0, "nm": 0, "refcohort": 0, "refcohort_fallback": 0, "cohort_bins": 0, "cohort_bins_rejected": 0}
while time.time() - t0 < budget:
    kind = rng.choice(["cohort", "cohort", "nm", "refcohort", "cohort_bins"])
    seed = int(rng.integers(1 << 30))
    if kind == "cohort":
        S = int(rng.choice([1, 2, 5, 63, 64, 65, 130, 257]))
        E = int(rng.integers(2, 3000)); C = int(rng.integers(1, 5))
        chrom_off, start, end = synth.exon_design(max(E, C), C, seed)
        E = int(chrom_off[-1])
        depth = float(rng.choice([5.0, 60.0, 300.0]))
        test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
        plan = ed.Plan(chrom_off, start, end)
        given = rng.random() < 0.3
        slab = int(rng.integers(1, S + 1)); nf = int(rng.choice([1, 2, 3, 4, 6, 8]))
        lanes = int(rng.choice([l for l in (1, 1, 2, 3, 4) if nf % l == 0] + ([0] if nf >= 4 else [])))     # (independent pipelines inside the object; 0 = slabs in flight / 2)
        layout = int(rng.integers(0, 2)); wire = int(rng.choice([2, 4]))
        if max(test.max(), ref.max()) >= 65536:
            wire = 4
        dt = np.int32 if wire == 4 else np.uint16
        th = test.astype(dt) if layout == 0 else np.ascontiguousarray(test.T.astype(dt))
        rh = ref.astype(dt) if layout == 0 else np.ascontiguousarray(ref.T.astype(dt))
        co = ed.Cohort(plan, slab, nf, own_queues=int(rng.integers(0, 2)), split=float(rng.choice([0.0, 0.3, 0.7])), lanes=lanes)
        out = co.run_host(th, rh, layout, phi=phi if given else None, expected=p if given else None, want_path=True)
        path = out["path"] if layout == 0 else out["path"].T
        b = ed.Batch(plan, S)
        if not given:
            # The fit is per sample, but the histogram geometry (= the order of summation) is picked per SLAB from its depth, so
            # a slab can differ from the whole batch in the last bits of (phi, expected): same maximum, 1e-10 apart at most.
            # Everything downstream is compared bit for bit GIVEN the pipeline's own parameters.
            dphi, dexp = ed.DeviceArray(np.zeros(S)), ed.DeviceArray(np.zeros(S))
            b.fit(test, ref, dphi, dexp)
            check(lib().ed_synchronize(None))
            fp, fe = dphi.to_host(), dexp.to_host()
            if not (np.allclose(out["phi"], fp, rtol=1e-9, atol=0) and np.allclose(out["expected"], fe, rtol=1e-10, atol=0)):
                raise SystemExit("cohort fit mismatch: seed %d S %d E %d slab %d" % (seed, S, E, slab))
        b.run(test, ref, out["phi"], out["expected"])
        ok = (out["calls"].tobytes() == b.calls().tobytes() and out["info"].tobytes() == b.call_info().tobytes() and np.array_equal(path, b.path()))
        if given:
            ok = ok and out["phi"].tobytes() == phi.tobytes() and out["expected"].tobytes() == p.tobytes()
        b.close()
        if not ok:
            np.savez("gpurun_out/fuzz_cohort_case.npz", test=test, ref=ref, chrom_off=chrom_off, start=start, end=end, slab=slab, nf=nf, layout=layout, wire=wire)
            raise SystemExit("cohort mismatch: seed %d S %d E %d slab %d in flight %d layout %d wire %d given %s" % (seed, S, E, slab, nf, layout, wire, given))
        co.close(); plan.close()
    elif kind == "cohort_bins":
        S = int(rng.choice([3, 5, 17, 64, 70])); C = int(rng.integers(1, 4)); E = int(rng.integers(600, 9000)); B = int(rng.integers(2, 9))
        chrom_off, start, end = synth.exon_design(E, C, seed)
        depth = float(rng.choice([30.0, 90.0, 250.0, 900.0]))          # (900: reference counts beyond the histogram form's bins)
        test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
        plan = ed.Plan(chrom_off, start, end)
        slab = int(rng.integers(2, S + 1)); nf = int(rng.integers(1, 4)); layout = int(rng.integers(0, 2))
        th = test if layout == 0 else np.ascontiguousarray(test.T)
        rh = ref if layout == 0 else np.ascontiguousarray(ref.T)
        co = ed.Cohort(plan, slab, nf, phi_bins=B)
        try:
            out = co.run_host(th, rh, layout, want_path=True)
        except ed.EdError as e:
            assert "Binning did not happen properly" in str(e), str(e)
            b = ed.Batch(plan, S)
            d = [ed.DeviceArray(np.zeros((B, S))), ed.DeviceArray(np.zeros((B + 1, S))), ed.DeviceArray(np.zeros(S))]
            try:
                b.fit_bins(test, ref, B, *d)
                raise SystemExit("cohort_bins: the pipeline rejects, the batch interface does not: seed %d" % seed)
            except ed.EdError:
                pass
            b.close(); co.close(); plan.close()
            n["cohort_bins_rejected"] += 1
            continue
        path = out["path"] if layout == 0 else out["path"].T
        b = ed.Batch(plan, S)
        d = [ed.DeviceArray(np.zeros((B, S))), ed.DeviceArray(np.zeros((B + 1, S))), ed.DeviceArray(np.zeros(S))]
        b.fit_bins(test, ref, B, *d)
        pb, eb, xb = [x.to_host() for x in d]
        if not (np.array_equal(out["edges"], eb) and np.allclose(out["phi_bins"], pb, rtol=1e-7, atol=0) and np.allclose(out["expected"], xb, rtol=1e-8, atol=0)):
            raise SystemExit("cohort_bins fit mismatch: seed %d S %d E %d B %d slab %d in flight %d depth %g" % (seed, S, E, B, slab, nf, depth))
        b.run_bins(test, ref, B, out["phi_bins"], out["edges"], out["expected"])
        if not (out["calls"].tobytes() == b.calls().tobytes() and out["info"].tobytes() == b.call_info().tobytes() and np.array_equal(path, b.path())):
            raise SystemExit("cohort_bins mismatch: seed %d S %d E %d B %d slab %d in flight %d layout %d depth %g" % (seed, S, E, B, slab, nf, layout, depth))
        b.close(); co.close(); plan.close()
    elif kind == "nm":
        S = int(rng.choice([1, 4, 9])); E = int(rng.integers(300, 20000))
        chrom_off, start, end = synth.exon_design(E, 1, seed)
        depth = float(rng.choice([20.0, 100.0, 600.0]))
        test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=0, mean_depth=depth)
        plan = ed.Plan(chrom_off, start, end)
        b = ed.Batch(plan, S)
        check(lib().ed_batch_set_fit_mode(b.handle, 1))
        dphi, dexp = ed.DeviceArray(np.zeros(S)), ed.DeviceArray(np.zeros(S))
        b.fit(test, ref, dphi, dexp)
        nu = b.fit_unconverged()[0]
        gp, ge = dphi.to_host(), dexp.to_host()
        for s in range(S):
            ophi, op, ne, fail = eo.fit_nm(test[:, s], ref[:, s], with_status=True)
            if fail:
                continue
            if not (abs(gp[s] - ophi) < 5e-3 * ophi and abs(ge[s] - op) < 5e-4 * op):
                raise SystemExit("aod-nm mismatch: seed %d sample %d device (%g, %g) checker (%g, %g) evals %d unconverged %d" % (seed, s, gp[s], ge[s], ophi, op, ne, nu))
        b.close(); plan.close()
    else:
        S = int(rng.choice([3, 8, 31, 33, 40])); E = int(rng.integers(400, 6000))
        lam = rng.lognormal(np.log(float(rng.choice([40.0, 120.0]))), 0.7, E)
        grp = rng.integers(0, 4, S)
        mu = lam[:, None] * rng.lognormal(0, 0.25, S)[None, :] * np.exp(rng.normal(0, 0.12, (E, 4))[:, grp] + rng.normal(0, 0.05, (E, S)))
        counts = rng.poisson(mu).astype(np.int32)
        bl = rng.integers(80, 600, E).astype(float) if rng.random() < 0.7 else None
        nred = int(rng.choice([0, 0, 300]))
        K = int(rng.choice([S - 1, 32, 4]))
        try:
            res = ed.cohort_select_reference_sets(counts, bl, nred, max_refs=K, want_reference=True)
        except ed.EdError as e:
            if "larger max_refs" in str(e) or "fewer than 2 bins" in str(e) or "no finite expected" in str(e):
                n["refcohort_fallback"] += 1
                continue
            raise
        agg = res["reference"].to_host().reshape(E, S)
        for t in range(S):
            others = np.delete(np.arange(S), t)
            one = ed.select_reference_set(counts[:, t], np.ascontiguousarray(counts[:, others]), bl, nred)
            want = [int(others[int(nm[1:]) - 1]) for nm in one["reference.choice"]]
            got = [int(v) for v in res["choice"][t, :res["n_chosen"][t]]]
            if got != want:
                # two candidates whose correlations agree to the last bits may order differently (Gram matrix vs two-pass sums)
                c1 = one["summary.stats"]["correlation"]
                close = np.min(np.abs(np.diff(c1[:max(len(want), len(got)) + 1]))) < 1e-12 if len(c1) > 1 else False
                if not close:
                    np.savez("gpurun_out/fuzz_refcohort_case.npz", counts=counts, bl=bl if bl is not None else np.zeros(0), nred=nred, K=K, t=t)
                    raise SystemExit("refcohort mismatch: seed %d S %d E %d t %d got %s want %s" % (seed, S, E, t, got, want))
            elif not np.array_equal(agg[:, t], counts[:, want].sum(axis=1)):
                raise SystemExit("aggregate reference mismatch: seed %d t %d" % (seed, t))
    n[kind] += 1
print("fuzz_cohort ok:", n, "%.0f s" % (time.time() - t0))
