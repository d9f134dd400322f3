import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import exomedepth_amd as ed
from exomedepth_amd import synth
E, S = 200_000, 1024
chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
torch.manual_seed(1)
test, ref, p, phi = synth.counts_torch(chrom_off, S, torch.device("cuda", 0), seed=20250623, mean_depth=100.0)
ts, rs = test.t().contiguous(), ref.t().contiguous()
plan = ed.Plan(chrom_off, start, end)
for ov in (1, 0):
    b = ed.Batch(plan, S); b.set_emit_mode(2); b.set_counts_layout(1); b.set_viterbi_overlap(ov)
    dphi, dexp = ed.DeviceArray(np.zeros(S)), ed.DeviceArray(np.zeros(S))
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.fit(ts, rs, dphi, dexp); b.run(ts, rs, dphi, dexp); n = b.n_calls()
        t1 = time.perf_counter()
        if rep >= 2: print("overlap groups", ov, "lone slab fit+run+n_calls: %.3f ms" % ((t1 - t0) * 1e3), n)
    b.close()
