"""Replay one case of fuzz_more.py's phi.bins mode: python tools/repro_bins.py E S B seed [C]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo, bins_oracle as bo

E, S, B, seed = (int(v) for v in sys.argv[1:5])
Cs = [int(sys.argv[5])] if len(sys.argv) > 5 else [1, 2, 3]
eo.build()
np.set_printoptions(linewidth=200)
for C in Cs:
    for depth in (30.0, 200.0):
        chrom_off, start, end = synth.exon_design(E, C, seed)
        test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
        plan = ed.Plan(chrom_off, start, end); batch = ed.Batch(plan, S)
        dphib = ed.DeviceArray(np.zeros((B, S))); dedges = ed.DeviceArray(np.zeros((B + 1, S))); dexp = ed.DeviceArray(np.zeros(S))
        try:
            batch.fit_bins(test, ref, B, dphib, dedges, dexp)
        except ed.EdError as e:
            print("C", C, "depth", depth, "rejected:", e); batch.close(); plan.close(); continue
        phib, ex = dphib.to_host(), dexp.to_host()
        batch.close(); plan.close()
        for s in range(min(S, 2)):
            ophi, op, _, _ = bo.fit_bins(test[:, s], ref[:, s], B)
            print("C", C, "depth", depth, "s", s, "\n dev", phib[:, s], ex[s], "\n ora", ophi, op)
