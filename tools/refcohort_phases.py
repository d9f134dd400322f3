"""Phase times of ed_cohort_select_reference_sets (ED_REFCOHORT_TIMING=1: the library synchronises and prints after every phase) on the
bench's workflow cohort: 200 000 exons x 1024 samples, n.bins.reduced 10 000, 32 candidates.    python tools/refcohort_phases.py [reps] [sm]"""
import os, sys, time
os.environ["ED_REFCOHORT_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
E, S = 200_000, 1024
chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
test, ref, p, phi = synth.counts_torch(chrom_off, S, torch.device("cuda", 0), seed=20250623, mean_depth=100.0)
bl = (np.asarray(end) - np.asarray(start)) / 1000.0
sm = len(sys.argv) > 2 and sys.argv[2] == "sm"
torch.manual_seed(20250623)
ref_t = torch.empty((S, E) if sm else (E, S), dtype=torch.int32, device=test.device)
cs = torch.empty((S, E), dtype=torch.int32, device=test.device) if sm else None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rs = ed.cohort_select_reference_sets(test, bl, 10000, max_refs=32, reference_out=ref_t, sample_major=sm, counts_sm_out=cs)
    torch.cuda.synchronize()
    print("[refcohort] TOTAL %.3f ms (with the per-phase synchronisations), mean chosen %.2f" % ((time.perf_counter() - t0) * 1e3, float(rs["n_chosen"].mean())), file=sys.stderr)
