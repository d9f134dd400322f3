"""Is the single-test select.reference.set entry reproducible run to run?  Sample 7465 of the 8 x 1 024-sample synthetic cohort against its 8 191 candidates,
five calls in one process: the rows must be identical bit for bit.    python tools/refset_determinism.py [sample]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
E, S, W = 200_000, 1024, 8
t = int(sys.argv[1]) if len(sys.argv) > 1 else 7465
dev = torch.device("cuda", 0)
if os.environ.get("SEED_GLOBAL", "1") == "1":
    torch.manual_seed(20250623)        # (synth draws its beta variates from torch's global generator: unseeded, the data differ from process to process)
chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
counts = torch.cat([synth.counts_torch(chrom_off, S, dev, seed=20250623 + r, mean_depth=100.0)[0] for r in range(W)], dim=1).contiguous()
print("data checksum", int(counts.to(torch.int64).sum().item()), int((counts.to(torch.int64) * torch.arange(S * W, device=dev)[None, :]).sum().item()))
bl = (np.asarray(end) - np.asarray(start)) / 1000.0
keep = [c for c in range(S * W) if c != t]
others = counts[:, keep].contiguous()
tc = counts[:, t].contiguous()
first = None
for rep in range(5):
    one = ed.select_reference_set(tc, others, bl, 10000, names=[str(c) for c in keep])
    rows = one["summary.stats"]
    nch = len(one["reference.choice"])
    if first is None:
        first = rows.copy()
        k = nch
        print("chosen", nch, "rows evaluated", int(np.sum(~np.isnan(rows["expected_BF"]))))
        for i in range(max(0, k - 4), k + 4):
            print(i + 1, {n: rows[n][i] for n in rows.dtype.names})
    else:
        same = all(np.array_equal(np.ascontiguousarray(first[n]).view(np.uint8), np.ascontiguousarray(rows[n]).view(np.uint8)) for n in rows.dtype.names)
        print("call", rep, "chosen", nch, "identical to the first call:", same)
        if not same:
            for n in rows.dtype.names:
                d = np.nonzero(first[n] != rows[n])[0]
                d = [i for i in d if not (isinstance(first[n][i], float) and np.isnan(first[n][i]) and np.isnan(rows[n][i]))]
                if len(d): print("   ", n, "differs at rows", d[:8], first[n][d[:3]], rows[n][d[:3]])
