"""Randomised sweep of ed_batch_fit_bins: the histogram form (csrc/edbins_hist.inc) and the per-cell form on the same counts --
complete.bins bit for bit between the two, the same rejections ("Binning did not happen properly"), per-level dispersions and the
expected proportion against the checker's long-double MLE (oracle/bins_oracle.py; up to three columns per case): the default form
at 1e-6 (a level on the device's floor phi = 1e-6 is compared as "both at most the floor": the checker follows an under-dispersed
level to ~1e-10), the per-cell form likewise (its largest difference is reported separately) -- over exon counts, slab widths, level counts and depths that cross every limit of the
histogram form (y >= 1024, n - lo >= 4096, r >= 8192, the 0.85 quantile beyond the r bins, lists that run out).
    python tools/fuzz_bins.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo, bins_oracle as bo
eo.build()

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
t0 = time.time()
n = {"cases": 0, "histogram_form": 0, "declined": 0, "rejected": 0, "columns_checked": 0}
worst = 0.0; worst_cell = 0.0
FLOOR = 1.0000001e-6
while time.time() - t0 < budget:
    seed = int(rng.integers(1 << 30))
    S = int(rng.choice([1, 3, 4, 5, 8, 13, 40])); C = int(rng.integers(1, 4)); B = int(rng.integers(2, 9))
    E = int(rng.choice([rng.integers(300, 3000), rng.integers(3000, 40000), rng.integers(40000, 140000)]))
    depth = float(rng.choice([12.0, 40.0, 90.0, 200.0, 450.0, 900.0]))
    chrom_off, start, end = synth.exon_design(E, C, seed)
    test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
    if rng.random() < 0.15:
        test = test * int(rng.integers(2, 20))                      # test counts beyond their bins
    plan = ed.Plan(chrom_off, start, end)
    res = []
    for form in (1, 0):
        batch = ed.Batch(plan, S)
        batch.set_fit_histograms(form)
        d = [ed.DeviceArray(np.zeros((B, S))), ed.DeviceArray(np.zeros((B + 1, S))), ed.DeviceArray(np.zeros(S))]
        try:
            batch.fit_bins(test, ref, B, *d)
            res.append(("ok", batch.fit_bins_form, [x.to_host() for x in d]))
        except ed.EdError as e:
            assert "Binning did not happen properly" in str(e), str(e)
            res.append(("rejected", None, None))
        batch.close()
    plan.close()
    case = (E, S, B, depth, seed)
    if os.environ.get("ED_FUZZ_VERBOSE"): print(case, res[0][0], res[0][1], round(time.time() - t0, 1), flush=True)
    assert res[0][0] == res[1][0], ("one form rejects, the other does not", case)
    n["cases"] += 1
    if res[0][0] == "rejected":
        n["rejected"] += 1
        continue
    n["histogram_form" if res[0][1] == 1 else "declined"] += 1
    (phib, edges, exp), (phib0, edges0, exp0) = res[0][2], res[1][2]
    assert np.array_equal(bits(edges), bits(edges0)), ("edges", case)
    for s in rng.permutation(S)[:3]:
        ophi, op, _, ocomplete = bo.fit_bins(test[:, s], ref[:, s], B)
        assert np.array_equal(bits(edges[:, s]), bits(ocomplete)), ("edges vs the checker", case, s)
        free = ~((phib[:, s] <= FLOOR) & (ophi <= FLOOR))
        err = max(np.max(np.abs(phib[free, s] - ophi[free]) / ophi[free], initial=0.0), abs(exp[s] - op) / op)
        free0 = ~((phib0[:, s] <= 3 * FLOOR) & (ophi <= 3 * FLOOR))
        err0 = max(np.max(np.abs(phib0[free0, s] - ophi[free0]) / ophi[free0], initial=0.0), abs(exp0[s] - op) / op)
        worst = max(worst, err); worst_cell = max(worst_cell, err0)
        n["columns_checked"] += 1
        # (holding a level at 1e-6 instead of the checker's ~0 moves the common intercept, and with it the other levels, by ~1e-6)
        assert err < (1e-6 if free.all() else 2e-5), ("default form against the checker", err, case, int(s), phib[:, s], ophi)
        assert err0 < (1e-6 if free0.all() else 2e-5), ("per-cell form against the checker", err0, case, int(s), phib0[:, s], ophi)
print("fuzz_bins: %s, largest relative difference from the checker: default form %.2e, per-cell form %.2e, %.0f s"
      % (n, worst, worst_cell, time.time() - t0))
