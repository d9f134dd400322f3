"""Which results differ from the batch interface, for which (slabs in flight, lanes, own_queues)?  (The flow of tests/test_gpu_cohort.py's first test.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
E, C, S = 12000, 5, 160
chrom_off, start, end = synth.exon_design(E, C, seed=11)
slabs = []
for k, n in enumerate((S, S, S, S, 70)):
    test, ref, _, _, _ = synth.counts_numpy(chrom_off, n, seed=300 + k, n_segments=4, mean_depth=80.0)
    slabs.append((test, ref))
plan = ed.Plan(chrom_off, start, end)
want = []
for test, ref in slabs:
    n = test.shape[1]
    b = ed.Batch(plan, n)
    dphi = ed.DeviceArray(np.zeros(n)); dexp = ed.DeviceArray(np.zeros(n))
    b.fit(test, ref, dphi, dexp); b.run(test, ref, dphi, dexp)
    want.append({"calls": b.calls().copy(), "info": b.call_info().copy(), "path": b.path().copy(), "loglik": b.loglik().copy(), "phi": dphi.to_host(), "expected": dexp.to_host()})
    b.close()
keys = ("calls", "info", "path", "loglik", "phi", "expected")
ORDER = [(int(a), int(b), int(c)) for a, b, c in (x.split(",") for x in sys.argv[1:])] or [(f, l, q) for f in (4, 6, 8) for l in (1, 0) for q in (1, 0)]
for in_flight, lanes, oq in ORDER:
    if True:
        if True:
            bad = {}
            for rep in range(int(os.environ.get('REPS', '3'))):
                co = ed.Cohort(plan, S, in_flight, timing=1, own_queues=oq, lanes=lanes)
                dev = [(ed.DeviceArray(t), ed.DeviceArray(r)) for t, r in slabs]
                tickets = []
                def check(j):
                    got = co.results(tickets[j], slabs[j % len(slabs)][0].shape[1], path=True, loglik=True)
                    for k in keys:
                        if got[k].tobytes() != want[j % len(slabs)][k].tobytes():
                            bad[k] = bad.get(k, 0) + 1
                for rounds in range(2):
                    for i, (dt, dr) in enumerate(dev):
                        if len(tickets) >= in_flight: check(len(tickets) - in_flight)
                        tickets.append(co.submit(dt, dr, n_samples=slabs[i][0].shape[1]))
                        if os.environ.get('SLEEP_MS'): import time; time.sleep(float(os.environ['SLEEP_MS']) * 1e-3)
                for j in range(len(tickets) - in_flight, len(tickets)): check(j)
                co.close()
            print("in flight", in_flight, "lanes", lanes or "auto", "own_queues", oq, "->", bad or "identical")
