"""Random cohorts through both forms of the cohort reference sets' chunk loop (csrc/edrefcohort.inc): the column-major kernel (k_rc_column: fit on
the tail counts of a column's histograms, median and RatioSd from the same bins; two geometries, hand-overs) against the row-major kernels of rounds 2-5
(ED_REFCOHORT_ROWMAJOR=1).  Same choices, the same medians, the other statistics to the fits' rounding.
    python tools/fuzz_refcohort.py [seconds] [seed] > profiles/rNN_fuzz_refcohort.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import exomedepth_amd as ed

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = cols = 0
forms = {"columns": 0, "row_major": 0}
large = beyond = 0
worst = {"phi": 0.0, "mean_p": 0.0, "ratio_sd": 0.0, "expected_BF": 0.0}
worst_flat = dict(worst)       # columns whose dispersion sits at the fit's lower bound (phi = 1e-6: numerically binomial, the likelihood flat in phi)
n_flat = 0
choice_diffs = []
while time.time() < t_end:
    E = int(rng.integers(3000, 30000)); S = int(rng.integers(8, 97)); depth = float(np.exp(rng.uniform(np.log(20), np.log(2000))))
    nred = int(rng.choice([0, 0, 2000, 5000])); K = int(min(S - 1, rng.choice([8, 16, 32])))
    lam = rng.lognormal(np.log(depth), rng.uniform(0.3, 0.9), E)
    sf = rng.lognormal(0, rng.uniform(0.05, 0.4), S)
    grp = rng.integers(0, 5, S)
    mu = lam[:, None] * sf[None, :] * np.exp(rng.normal(0, 0.12, (E, 5))[:, grp] + rng.normal(0, rng.uniform(0.02, 0.15), (E, S)))
    counts = rng.poisson(mu).astype(np.int32)
    bl = rng.integers(80, 600, E).astype(float)
    try:
        a = ed.cohort_select_reference_sets(counts, bl, nred, max_refs=K, want_reference=False)
    except ed.EdError as e:          # (fewer than 2 bins selected and the like: the same for both forms)
        continue
    path = ed.refcohort_last_path()
    os.environ["ED_REFCOHORT_ROWMAJOR"] = "1"
    b = ed.cohort_select_reference_sets(counts, bl, nred, max_refs=K, want_reference=False)
    del os.environ["ED_REFCOHORT_ROWMAJOR"]
    cases += 1
    forms["columns" if path["chunks_by_columns"] else "row_major"] += 1
    large += path["columns_large_geometry"]; beyond += path["columns_beyond_bins"]
    ra, rb = a["summary.stats"], b["summary.stats"]
    assert a["n.bins"] == b["n.bins"] and np.array_equal(ra["ref_index"], rb["ref_index"])
    reached = ~np.isnan(ra["expected_BF"]) & ~np.isnan(rb["expected_BF"])
    cols += int(reached.sum())
    assert np.array_equal(ra["median_depth"][reached], rb["median_depth"][reached]), "median"
    flat = reached & ((ra["phi"] < 2e-6) | (rb["phi"] < 2e-6))
    n_flat += int(flat.sum())
    for f in worst:
        for mask, acc in ((reached & ~flat, worst), (flat, worst_flat)):
            x, y = ra[f][mask], rb[f][mask]
            if x.size:
                d = np.abs(x - y) / np.maximum(np.abs(y), 1e-300)
                if f == ("phi" if acc is worst else "mean_p") and d.max() > 1e-7 and d.max() > acc[f]:      # keep the case: which form is off?  (tools/fuzz_refcohort_case.py: against the checker's MLE)
                    t, i = np.argwhere(mask)[int(np.argmax(d))]
                    np.savez_compressed("gpurun_out/fuzz_refcohort_case%s.npz" % ("_flat" if acc is worst_flat else ""), counts=counts, bl=bl, nred=nred, K=K, t=t, i=i,
                                        cols=np.array([ra["mean_p"][t, i], rb["mean_p"][t, i]]), phi=np.array([ra["phi"][t, i], rb["phi"][t, i]]))
                acc[f] = max(acc[f], float(d.max()))
    if not (np.array_equal(a["n_chosen"], b["n_chosen"]) and np.array_equal(a["choice"], b["choice"])):
        for t in np.where(a["n_chosen"] != b["n_chosen"])[0]:
            bf = rb["expected_BF"][t]
            i, j = int(a["n_chosen"][t]) - 1, int(b["n_chosen"][t]) - 1
            choice_diffs.append((E, S, round(depth, 1), int(t), i + 1, j + 1, float(abs(bf[i] - bf[j]) / abs(bf[j]))))
print("cases %d (E 3000-30000, S 8-96, depth 20-2000, n.bins.reduced 0 / 2000 / 5000, max_refs 8 / 16 / 32), cumulative references compared %d" % (cases, cols))
print("chunks served by the column-major kernel %d, by the row-major kernels (too deep) %d; columns by the large geometry %d, columns with values beyond their bins %d"
      % (forms["columns"], forms["row_major"], large, beyond))
print("medians: identical.  Largest relative difference column-major vs row-major: " + ", ".join("%s %.2e" % kv for kv in worst.items()))
print("... on the %d columns with the dispersion at its lower bound (phi = 1e-6; the row-major passes stop short there, see tools/fuzz_refcohort_case.py): " % n_flat
      + ", ".join("%s %.2e" % kv for kv in worst_flat.items()))
print("tests whose choice differs between the forms: %d" % len(choice_diffs))
for c in choice_diffs[:20]:
    print("   E %d S %d depth %s test %d: %d vs %d references, expected.BF of the two candidates %.1e apart (relative)" % c)
