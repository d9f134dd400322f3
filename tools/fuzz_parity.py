"""Randomised parity sweep of the batched pipeline against the CPU checker (tests/ holds the fixed cases; this
is the wide net: odd shapes, sample counts around the tile / XCD-numbering boundaries, tiny and empty chromosomes,
extreme dispersions, tumour mixtures, cells without reads, counts beyond the tables).
    python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (initialises the HIP runtime first)
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
t0 = time.time()
n_cases = n_cells = n_fits = 0
while time.time() - t0 < budget:
    S = int(rng.choice([1, 3, 16, 63, 64, 65, 127, 512, 513, 520, 576, 640, 1000]))
    E = int(rng.integers(1, 40 if S > 400 else 600) * rng.choice([1, 7]))
    C = int(rng.integers(1, 6))
    seed = int(rng.integers(1 << 30))
    chrom_off, start, end = synth.exon_design(max(E, C), C, seed)
    E = int(chrom_off[-1])
    if rng.random() < 0.3 and C > 1:               # an empty chromosome in the middle
        k = int(rng.integers(1, C))
        chrom_off = np.insert(chrom_off, k, chrom_off[k]).astype(np.int32)
        C += 1
    depth = float(rng.choice([3.0, 40.0, 150.0, 2500.0]))
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
    if rng.random() < 0.3:
        test, ref, p = ref.copy(), test.copy(), 1.0 - p
    if rng.random() < 0.5:
        dead = rng.random(test.shape) < 0.2
        test[dead] = 0; ref[dead] = 0
    phi = phi * float(rng.choice([1.0, 1e-3, 30.0]))
    phi = np.minimum(phi, 0.6)
    special = []
    if rng.random() < 0.15:                        # a few samples outside the model's domain: phi >= 1, expected 0 / 1 / beyond
        k = int(rng.integers(1, max(2, S // 8 + 1)))
        idx = rng.choice(S, size=min(k, S), replace=False)
        special = [int(j) for j in idx[:4]]
        phi = phi.copy(); p = p.copy()
        for j in idx:
            mode = int(rng.integers(0, 4))
            if mode == 0: phi[j] = float(rng.uniform(1.0, 4.0))
            elif mode == 1: p[j] = float(rng.choice([0.0, 1.0]))
            elif mode == 2: p[j] = float(rng.uniform(1.0, 1.5))
            else: phi[j] = float(rng.choice([0.0, 1.0]))
    mixture = float(rng.choice([1.0, 1.0, 0.4]))
    plan = ed.Plan(chrom_off, start, end, float(rng.choice([1e-4, 1e-2])), float(rng.choice([5e4, 2e3])))
    batch = ed.Batch(plan, S)
    batch.run(test, ref, phi, p, mixture=mixture)
    ll, path, calls = batch.loglik(), batch.path(), batch.calls()
    tp, L = plan.transition_probability, plan.expected_CNV_length
    batch.close(); plan.close()
    for s in list(rng.choice(S, size=min(S, 6), replace=False)) + special:
        ell, _ = eo.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], mixture, eo.PORTABLE)
        if not np.all((bits(ll[:, :, s]) == bits(ell)) | (np.isnan(ll[:, :, s]) & np.isnan(ell))):
            bad = np.argwhere(bits(ll[:, :, s]) != bits(ell))
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez_compressed("gpurun_out/fuzz_loglik_case.npz", chrom_off=chrom_off, start=start, end=end, test=test, ref=ref, p=p, phi=phi,
                                mixture=mixture, tp=tp, L=L, ll=ll, ell=ell, s=s)
            for e, st in bad[:10]:
                print("  case", n_cases, "exon", e, "state", st, "obs", test[e, s], "tot", test[e, s] + ref[e, s], "dev %r" % ll[e, st, s], "ora %r" % ell[e, st])
            # once more, same inputs, fresh batch: does the device repeat itself?
            plan = ed.Plan(chrom_off, start, end, tp, L); batch = ed.Batch(plan, S)
            batch.run(test, ref, phi, p, mixture=mixture)
            ll2 = batch.loglik(); batch.close(); plan.close()
            print("  re-run equals first run:", np.array_equal(bits(ll2), bits(ll)), " re-run equals checker:", np.array_equal(bits(ll2[:, :, s]), bits(ell)))
            raise AssertionError(("loglik", E, S, C, seed, int(s), len(bad)))
        epath, ecalls = eo.callcnvs(ell, chrom_off, start, end, tp, L)
        assert np.array_equal(path[:, s].astype(np.int8), epath), ("path", E, S, C, seed, s)
        mine = calls[calls["sample"] == s]
        assert len(mine) == len(ecalls) and np.array_equal(mine["start_exon"] + 1, ecalls[:, 0].astype(np.int64)), ("calls", E, S, C, seed, s)
    if E >= 200 and rng.random() < 0.5:            # the dispersion fit (histogram form) against the checker's MLE
        plan = ed.Plan(chrom_off, start, end)
        batch = ed.Batch(plan, S)
        dphi = ed.DeviceArray(np.zeros(S)); dexp = ed.DeviceArray(np.zeros(S))
        batch.fit(test, ref, dphi, dexp)
        batch.run(test, ref, dphi, dexp)           # (also exercises fit -> run on device-resident parameters)
        batch.n_calls()
        fphi, fexp = dphi.to_host(), dexp.to_host()
        batch.close(); plan.close()
        for s in rng.choice(S, size=min(S, 3), replace=False):
            ophi, op, _, it = eo.fit_mle(test[:, s], ref[:, s])
            if it >= 0 and 1e-5 < ophi < 0.5:
                # binary64 digamma against the checker's long double: the gradient's cancellation grows with a + b = 1/phi
                tol_phi = max(1e-7, 1e-13 / ophi ** 2)   # binary64 resolves the maximum to ~2e-14 / phi^2
                if not (abs(fphi[s] - ophi) < tol_phi * ophi and abs(fexp[s] - op) < 1e-7 * op):
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez_compressed("gpurun_out/fuzz_fit_case.npz", test=test[:, s], ref=ref[:, s], fphi=fphi[s], fexp=fexp[s], ophi=ophi, op=op)
                    raise AssertionError(("fit", E, S, seed, int(s), fphi[s], ophi, fexp[s], op))
                n_fits += 1
    n_cases += 1; n_cells += E * S
print("fuzz ok: %d cases, %d cells, %d fits, %.0f s" % (n_cases, n_cells, n_fits, time.time() - t0))
