"""One rank's share of an 8-rank sample-sharded cohort (what bench.py --gpus 8 runs in its workflow leg on every rank, after the all-gather of the count
slabs): 200 000 bins x 8 192 samples on the device, the rank's own 1 024 samples as tests, all 8 192 as candidates.  Checks that the call goes through at
that geometry, that spot tests equal the single-test entry's choice, and prints the time.    python tools/check_refsets_8192.py [rank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
E, S, W = 200_000, 1024, 8
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
torch.manual_seed(20250623)            # (synth draws its beta variates from torch's global generator)
chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
cols = [synth.counts_torch(chrom_off, S, dev, seed=20250623 + r, mean_depth=100.0)[0] for r in range(W)]
counts = torch.cat(cols, dim=1).contiguous()
del cols
bl = (np.asarray(end) - np.asarray(start)) / 1000.0
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rs = ed.cohort_select_reference_sets(counts, bl, 10000, max_refs=32, test_range=(rank * S, (rank + 1) * S))
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("tests %d..%d of %d: %.1f ms, mean chosen %.2f, reference %s" % (rank * S, (rank + 1) * S, S * W, (t1 - t0) * 1e3, float(rs["n_chosen"].mean()), rs["reference"].shape))
bad = 0
for t in (rank * S, rank * S + 517, (rank + 1) * S - 1):
    keep = [c for c in range(S * W) if c != t]
    one = ed.select_reference_set(counts[:, t].contiguous(), counts[:, keep].contiguous(), bl, 10000, names=[str(c) for c in keep])
    mine = rs["choice"][t - rank * S][: rs["n_chosen"][t - rank * S]]
    theirs = np.asarray([int(c) for c in one["reference.choice"]])
    ok = np.array_equal(mine, theirs)
    bad += 0 if ok else 1
    print("test", t, "chosen", len(mine), "equal to the single-test entry (same columns, same order):", ok)
print("ok" if bad == 0 else "MISMATCH")
