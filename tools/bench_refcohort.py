"""Timing of the cohort-level select.reference.set (ed_cohort_select_reference_sets) next to single-test calls.
    python tools/bench_refcohort.py [E] [S] [n_bins_reduced] [single calls to time]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth

E = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nred = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
nsingle = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda", 0)
chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=20250620 + 3)
counts = test.contiguous()           # the cohort's own counts: every sample a test, the others its candidates
bl = (end - start) / 1000.0
torch.cuda.synchronize()
out = {}
for rep in range(2):
    t0 = time.perf_counter()
    res = ed.cohort_select_reference_sets(counts, bl, nred, max_refs=32)
    torch.cuda.synchronize()
    out["cohort_s"] = time.perf_counter() - t0
out["n_chosen_mean"] = float(res["n_chosen"].mean()); out["n_bins"] = res["n.bins"]
ts = []
for t in range(nsingle):
    others = torch.cat([counts[:, :t], counts[:, t + 1:]], dim=1).contiguous()
    tt = counts[:, t].contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one = ed.select_reference_set(tt, others, bl, nred)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
    idx = [i + (1 if i >= t else 0) for i in [int(n[1:]) - 1 for n in one["reference.choice"]]]
    assert idx == [int(v) for v in res["choice"][t, :res["n_chosen"][t]]], (t, idx, res["choice"][t])
out["single_call_s_median"] = float(np.median(ts)); out["single_calls_timed"] = nsingle
out["all_single_calls_extrapolated_s"] = out["single_call_s_median"] * S
out["speedup"] = out["all_single_calls_extrapolated_s"] / out["cohort_s"]
out["workload"] = "%d bins x %d samples, n.bins.reduced = %d" % (E, S, nred)
print(json.dumps(out))
