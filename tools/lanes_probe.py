"""Two independent cohort pipelines in ONE process (slabs dealt alternately) against one pipeline: ms per 200 000 x 1024 slab, fit on, table mode, sample-major counts.
Two PROCESSES sharing the GPU run at 3.8 - 3.9 ms per slab against 4.15 for one (tools/ab_two_procs.sh): is it the second pipeline or the second process?
    python tools/lanes_probe.py [slots per pipeline]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
E, S = 200_000, 1024
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
torch.manual_seed(20250623)
chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=20250623, mean_depth=100.0)
ts, rs = test.t().contiguous(), ref.t().contiguous()
plan = ed.Plan(chrom_off, start, end, 1e-4, 50000.0)
def run(n_lanes, steps=40, warm=8):
    cos = [ed.Cohort(plan, S, slots, emit_mode=2, counts_layout=1) for _ in range(n_lanes)]
    for i in range(warm): cos[i % n_lanes].submit(ts, rs, n_samples=S)
    for c in cos: c.drain()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): cos[i % n_lanes].submit(ts, rs, n_samples=S)
    for c in cos: c.drain()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    for c in cos: c.close()
    return dt
for rep in range(3):
    print("pipelines 1: %.3f ms per slab   2: %.3f   3: %.3f" % (run(1), run(2), run(3)))
