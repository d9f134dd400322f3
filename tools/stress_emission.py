"""Run the same small batches again and again in one process and compare every likelihood matrix with the first one:
looks for a rare device-side race (see tools/repro_emission.py).  python tools/stress_emission.py [seconds]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(root, 'tests/golden/emission_case_small_shapes.npz'))
cases = [(d['chrom_off'], d['start'], d['end'], d['test'], d['ref'], d['phi'], d['p'], float(d['tp']), float(d['L']))]
for k, (E, S, depth, mult) in enumerate([(300, 65, 3.0, 30.0), (40, 513, 3.0, 30.0), (500, 3, 40.0, 1.0), (150, 127, 2500.0, 1e-3)]):
    co, st, en = synth.exon_design(E, 2, 100 + k)
    t, r, p, phi, _ = synth.counts_numpy(co, S, 100 + k, n_segments=2, mean_depth=depth)
    cases.append((co, st, en, t, r, np.minimum(phi * mult, 0.6), p, 1e-2, 2e3))
state = []
for (co, st, en, t, r, phi, p, tp, L) in cases:
    plan = ed.Plan(co, st, en, tp, L); b = ed.Batch(plan, t.shape[1])
    b.run(t, r, phi, p); first = b.loglik().copy(); fpath = b.path().copy()
    state.append((plan, b, first, fpath))
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    for k, (co, st, en, t, r, phi, p, tp, L) in enumerate(cases):
        plan, b, first, fpath = state[k]
        b.run(t, r, phi, p)
        ll = b.loglik(); path = b.path()
        if not np.array_equal(bits(ll), bits(first)) or not np.array_equal(path, fpath):
            bad = np.argwhere(bits(ll) != bits(first))
            print("run", n, "case", k, "differs from its first run in", len(bad), "values;", bad[:10].tolist())
            for e, s_, q in bad[:10]:
                print("   exon", e, "state", s_, "sample", q, "obs", t[e, q], "tot", t[e, q] + r[e, q], "now %r" % ll[e, s_, q], "first %r" % first[e, s_, q])
            sys.exit(1)
        n += 1
print("stable:", n, "runs of", len(cases), "batches in %.0f s" % (time.time() - t0))
