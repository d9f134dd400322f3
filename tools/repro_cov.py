"""Replay one case of fuzz_more.py's covariate mode: python tools/repro_cov.py E S K seed [lam0]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exomedepth_amd as ed
from oracle import edoracle as eo
from exomedepth_amd import synth

E, S, K, seed = (int(v) for v in sys.argv[1:5])
lams = [float(sys.argv[5])] if len(sys.argv) > 5 else [40.0, 200.0]
eo.build()
for lam0 in lams:
    chrom_off, start, end = synth.exon_design(E, 1, seed)
    r2 = np.random.default_rng(seed)
    X = np.ascontiguousarray(np.stack([r2.uniform(-0.2, 0.2, E), r2.normal(0, 1, E), r2.uniform(-1, 1, E)], axis=1)[:, :K])
    lam = r2.lognormal(np.log(lam0), 0.6, E)
    test = np.zeros((E, S), dtype=np.int32); ref = np.zeros((E, S), dtype=np.int32)
    for s in range(S):
        beta = np.concatenate([[r2.uniform(-2.4, -1.6)], r2.uniform(-0.8, 0.8, K) * np.array([2.0, 0.15, 0.3])[:K]])
        phi_t = r2.uniform(0.003, 0.012)
        pe = 1 / (1 + np.exp(-(beta[0] + X @ beta[1:])))
        tot = r2.poisson(lam * 9)
        yy = r2.binomial(tot, r2.beta(pe * (1 - phi_t) / phi_t, (1 - pe) * (1 - phi_t) / phi_t))
        test[:, s] = yy; ref[:, s] = tot - yy
    plan = ed.Plan(chrom_off, start, end)
    for cols in [list(range(S))] + [[s] for s in range(S)]:
        batch = ed.Batch(plan, len(cols))
        dbeta = ed.DeviceArray(np.zeros((K + 1, len(cols)))); dphi = ed.DeviceArray(np.zeros(len(cols)))
        batch.fit_cov(np.ascontiguousarray(test[:, cols]), np.ascontiguousarray(ref[:, cols]), X, dbeta, dphi)
        print("lam0", lam0, "cols", cols, "\n dev beta", dbeta.to_host().T, "phi", dphi.to_host())
        batch.close()
    for s in range(S):
        print(" oracle", s, eo.fit_mle_cov(test[:, s], ref[:, s], X))
    plan.close()
