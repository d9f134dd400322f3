"""Replay tests/golden/emission_case_small_shapes.npz (a case tools/fuzz_parity.py failed on ONCE, 13 013 cases into a
run) many times, interleaved with other batches that disturb the allocator, and tell which side moves."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo
eo.build()
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/emission_case_small_shapes.npz'))
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
rng = np.random.default_rng(1)
ref_dev = ref_ora = None
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for it in range(n_iter):
    plan = ed.Plan(d['chrom_off'], d['start'], d['end'], float(d['tp']), float(d['L']))
    b = ed.Batch(plan, 1)
    b.run(d['test'], d['ref'], d['phi'], d['p'], mixture=float(d['mixture']))
    ll = b.loglik()[:, :, 0].copy()
    b.close(); plan.close()
    ell, _ = eo.get_loglike_matrix(d['phi'][0], d['p'][0], d['test'][:, 0] + d['ref'][:, 0], d['test'][:, 0], float(d['mixture']), eo.PORTABLE)
    if ref_dev is None: ref_dev, ref_ora = ll.copy(), ell.copy()
    dev_moved = not np.array_equal(bits(ll), bits(ref_dev)); ora_moved = not np.array_equal(bits(ell), bits(ref_ora))
    if dev_moved or ora_moved or not np.array_equal(bits(ll), bits(ell)):
        bad = np.argwhere(bits(ll) != bits(ell))
        print("iteration", it, "device moved", dev_moved, "checker moved", ora_moved, "mismatches", len(bad))
        for e, st in bad[:8]:
            print("  exon", e, "state", st, "obs", d['test'][e, 0], "tot", d['test'][e, 0] + d['ref'][e, 0], "dev %r" % ll[e, st], "ora %r" % ell[e, st], "first dev %r" % ref_dev[e, st])
        sys.exit(1)
    # disturb: another batch of random shape
    S2 = int(rng.choice([1, 3, 64, 65, 513])); E2 = int(rng.integers(1, 300)); C2 = int(rng.integers(1, 4))
    co, st_, en = synth.exon_design(max(E2, C2), C2, it)
    t2, r2, p2, phi2, _ = synth.counts_numpy(co, S2, it, n_segments=2, mean_depth=float(rng.choice([3.0, 150.0, 2500.0])))
    pl2 = ed.Plan(co, st_, en); b2 = ed.Batch(pl2, S2)
    if it % 3 == 0 and int(co[-1]) >= 200:
        dphi = ed.DeviceArray(np.zeros(S2)); dexp = ed.DeviceArray(np.zeros(S2))
        b2.fit(t2, r2, dphi, dexp); b2.run(t2, r2, dphi, dexp); b2.n_calls()
    else:
        b2.run(t2, r2, phi2, p2); b2.calls()
    b2.close(); pl2.close()
print("stable over", n_iter, "iterations")
