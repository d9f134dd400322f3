"""Dispersion-fit time against read depth (share of counts beyond the first-level histogram bins), both count layouts, with the fit's
kernels timed by rocprofv3 when run under it:  python tools/fit_timing.py [layout ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exomedepth_amd as ed
from exomedepth_amd import synth

E, S = 200_000, 1024
dev = torch.device("cuda:0")
chrom_off, start, end = synth.exon_design(E, 24, 20250623)
plan = ed.Plan(chrom_off, start, end)
layouts = [int(a) for a in sys.argv[1:]] or [0, 1]
for depth in (25.0, 50.0, 100.0, 200.0):
    test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=7, mean_depth=depth)
    over = float(((ref >= 4096) | (test + ref >= 4096) | (test >= 1024)).double().mean().item())
    for layout in layouts:
        batch = ed.Batch(plan, S)
        t_in, r_in = (test.t().contiguous(), ref.t().contiguous()) if layout else (test, ref)
        if layout:
            batch.set_emit_mode(2); batch.set_counts_layout(1)
        dphi = torch.zeros(S, dtype=torch.float64, device=dev); dexp = torch.zeros(S, dtype=torch.float64, device=dev)
        for _ in range(2):
            batch.fit(t_in, r_in, dphi, dexp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            batch.fit(t_in, r_in, dphi, dexp)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        err = float(((dphi - phi).abs() / phi).median().item())
        print("depth %6.1f  layout %d  beyond the first-level bins %.4f  fit %.3f ms  median |phi - planted|/planted %.3f  unconverged %d"
              % (depth, layout, over, ms, err, batch.fit_unconverged()[0]))
        batch.close()
    del test, ref
plan.close()
