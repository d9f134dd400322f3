"""One-off: BASELINE configs[3]'s whole cohort (200 000 exons x 8192 samples, 1.6e9 cells) on ONE GPU -- index
arithmetic beyond 2^31 cells, ~70 GB of HBM -- with whole columns checked against the CPU checker."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo

E, S, C = 200_000, 8192, 24
dev = torch.device("cuda", 0)
chrom_off, start, end = synth.exon_design(E, C, seed=20250620)
test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=20250699)
plan = ed.Plan(chrom_off, start, end)
batch = ed.Batch(plan, S)
phi_f = torch.empty(S, dtype=torch.float64, device=dev); p_f = torch.empty(S, dtype=torch.float64, device=dev)
batch.fit(test, ref, phi_f, p_f)
batch.run(test, ref, phi_f, p_f)
calls = batch.calls()
phi_h, p_h = phi_f.cpu().numpy(), p_f.cpu().numpy()
for s in (0, 4095, 4096, 8191):
    t = test[:, s].cpu().numpy(); r = ref[:, s].cpu().numpy()
    ophi, op, _, _ = eo.fit_mle(t, r)
    assert abs(phi_h[s] - ophi) < 1e-7 * ophi and abs(p_h[s] - op) < 1e-8 * op, ("fit", s)
    ell, _ = eo.get_loglike_matrix(phi_h[s], p_h[s], t + r, t, 1.0, eo.PORTABLE)
    epath, ecalls = eo.callcnvs(ell, chrom_off, start, end)
    mine = calls[calls["sample"] == s]
    assert len(mine) == len(ecalls) and np.array_equal(mine["start_exon"] + 1, ecalls[:, 0].astype(np.int64)) \
        and np.array_equal(mine["end_exon"] + 1, ecalls[:, 1].astype(np.int64)) and np.array_equal(mine["nexons"], ecalls[:, 3].astype(np.int64)), ("calls", s)
path = batch.path()
for s in (0, 4095, 4096, 8191):
    t = test[:, s].cpu().numpy(); r = ref[:, s].cpu().numpy()
    ell, _ = eo.get_loglike_matrix(phi_h[s], p_h[s], t + r, t, 1.0, eo.PORTABLE)
    epath, _ = eo.callcnvs(ell, chrom_off, start, end)
    assert np.array_equal(path[:, s].astype(np.int8), epath), ("path", s)
print("200000 x 8192 on one GPU: fit, calls and paths of 4 whole columns match the checker; %d calls" % len(calls))
batch.close(); plan.close()
