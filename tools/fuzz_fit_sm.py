"""Randomised sweep of the dispersion fit from sample-major counts (k_fit_moments_sm / k_fit_hist_sm / k_fit_hnewton on a sample's own row of
bins) against the same fit from [exons][samples] counts (k_fit_hist's quads, the lists re-binned by the Newton kernel) and, on spot columns,
against the checker's long-double maximum-likelihood fit: shapes from 1 exon to 250 000, depths 2 .. 4000 (all three histogram geometries),
columns of identical counts (bins that leave the LDS early), zeros, subset.for.speed steps.
    python tools/fuzz_fit_sm.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260929)
t0 = time.time()
n_cases = n_cols = n_oracle = 0
worst = worst_o = 0.0
while time.time() - t0 < budget:
    big = rng.random() < 0.15
    S = int(rng.choice([1, 2, 3, 4, 5, 8, 17, 64])) if big else int(rng.choice([1, 3, 4, 7, 16, 33, 130, 260]))
    E = int(rng.integers(60_000, 250_000)) if big else int(rng.integers(1, 3000) * rng.choice([1, 9]))
    C = int(rng.integers(1, 5))
    seed = int(rng.integers(1 << 30))
    chrom_off, start, end = synth.exon_design(max(E, C), C, seed)
    E = int(chrom_off[-1])
    depth = float(np.exp(rng.uniform(np.log(2.0), np.log(4000.0))))
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
    kind = rng.random()
    if kind < 0.15:                                   # a few values only: hot bins
        test[:, :] = rng.integers(0, 3, size=test.shape) * int(rng.integers(1, 50)) + int(rng.integers(0, 30))
        ref[:, :] = rng.integers(0, 2, size=ref.shape) * int(rng.integers(1, 500)) + int(rng.integers(1, 900))
    elif kind < 0.3:
        dead = rng.random(test.shape) < 0.3
        test[dead] = 0; ref[dead] = 0
    by = int(rng.choice([1, 1, 1, 2, 7])) if E > 50 else 1
    plan = ed.Plan(chrom_off, start, end)
    out = []
    for layout in (0, 1):
        b = ed.Batch(plan, S)
        if layout:
            b.set_emit_mode(2); b.set_counts_layout(1)
        t_in, r_in = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
        dt, dr = ed.DeviceArray(t_in), ed.DeviceArray(r_in)
        dphi, dexp = ed.DeviceArray(np.zeros(S)), ed.DeviceArray(np.zeros(S))
        for _ in range(int(rng.integers(1, 3))):
            b.fit(dt, dr, dphi, dexp, by=by)
        out.append((dphi.to_host(), dexp.to_host(), b.fit_unconverged()[0]))
        b.close()
    (p0, e0, u0), (p1, e1, u1) = out
    ok = np.isfinite(p0) & np.isfinite(p1) & (p0 > 0)
    assert np.array_equal(np.isfinite(p0), np.isfinite(p1)), ("finite", E, S, seed, depth)
    # the same maximum from two orders of summation: the Newton tolerance (1e-9 on the step) bounds the difference
    rel = np.abs(p1[ok] - p0[ok]) / p0[ok]
    if rel.size:
        if u0 == 0 and u1 == 0 and E // by >= 40:      # (a handful of rows: a flat likelihood, where two orders of summation may part ways)
            assert rel.max() < 2e-6, ("phi", E, S, seed, depth, by, float(rel.max()))
            assert np.max(np.abs(e1[ok] - e0[ok]) / e0[ok]) < 1e-7, ("expected", E, S, seed, depth)
            worst = max(worst, float(rel.max()))
    if 40 <= E <= 30_000 and by == 1:
        for s in rng.choice(S, size=min(S, 2), replace=False):
            if not ok[s] or test[:, s].sum() == 0:
                continue
            op, oe = eo.fit_mle(test[:, s], ref[:, s])[:2]
            if np.isfinite(op) and 1e-7 < op < 0.5 and u1 == 0:
                d = abs(p1[s] - op) / op
                assert d < 1e-4, ("oracle", E, S, seed, depth, int(s), p1[s], op)
                worst_o = max(worst_o, d); n_oracle += 1
    plan.close()
    n_cases += 1; n_cols += S
print("fuzz_fit_sm ok: %d cases, %d columns, sample-major fit against the [E][S] fit (columns of >= 40 rows, both converged): max relative difference in phi %.2e; %d columns against the "
      "checker's long-double MLE: max %.2e; %.0f s" % (n_cases, n_cols, worst, n_oracle, worst_o, time.time() - t0))
