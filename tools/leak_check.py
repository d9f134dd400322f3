"""Device memory before/after repeated plan/batch create, fit, run, phi.bins, close cycles: python tools/leak_check.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exomedepth_amd as ed
from exomedepth_amd import synth
dev = torch.device("cuda:0")
E, S = 20000, 256
chrom_off, start, end = synth.exon_design(E, 6, 5)
test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=5)
torch.cuda.synchronize()
def free(): return torch.cuda.mem_get_info()[0]
f0 = None
for it in range(60):
    plan = ed.Plan(chrom_off, start, end); b = ed.Batch(plan, S)
    dphi = torch.zeros(S, dtype=torch.float64, device=dev); dexp = torch.zeros(S, dtype=torch.float64, device=dev)
    b.fit(test, ref, dphi, dexp); b.run(test, ref, dphi, dexp); n = len(b.calls())
    if it % 3 == 0:
        B = 3; dphib = torch.zeros((B, S), dtype=torch.float64, device=dev); ded = torch.zeros((B + 1, S), dtype=torch.float64, device=dev)
        b.fit_bins(test, ref, B, dphib, ded, dexp); b.run_bins(test, ref, B, dphib, ded, dexp)
    b.close(); plan.close()
    torch.cuda.synchronize()
    if it == 4: f0 = free()
print("free after 5 cycles %d MB, after 60 cycles %d MB, calls %d" % (f0 >> 20, free() >> 20, n))
