"""fit_concordance (exomedepth_amd/concordance.py: the whole path with fit mode 0 against fit mode 1) over ALL columns of the
bench's configs[2] batch, in chunks of columns, aggregated:  python tools/fit_concordance_full.py [out.json] [columns] [chunk]"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import concordance, synth

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fit_concordance_full.json"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 128
E, C = 200_000, 24
chrom_off, start, end = synth.exon_design(E, C, seed=20250620 + 3)
test, ref, p, phi = synth.counts_torch(chrom_off, S, torch.device("cuda:0"), seed=20250620 + 3, mean_depth=100.0)
plan = ed.Plan(chrom_off, start, end, 1e-4, 50000.0)
agg = None
for s0 in range(0, S, chunk):
    r = concordance.fit_mode_concordance(plan, test[:, s0:s0 + chunk].contiguous(), ref[:, s0:s0 + chunk].contiguous())
    if agg is None:
        agg = dict(r)
    else:
        for k in ("columns", "cells", "discordant_states", "columns_with_discordant_states", "calls_mle", "calls_aod_nm", "discordant_call_rows",
                  "unconverged_mle", "unconverged_aod_nm"):
            agg[k] += r[k]
        for k in ("max_rel_dphi", "max_rel_dexpected", "max_rel_dloglik"):
            agg[k] = max(agg[k], r[k])
        agg["median_rel_dphi"] = None
    print(s0, r["discordant_states"], r["discordant_call_rows"], flush=True)
agg["workload"] = "bench.py's configs[2] batch (seed 20250620 + 3): %d exons x %d samples, all columns, chunks of %d" % (E, S, chunk)
json.dump(agg, open(out, "w"), indent=1)
print(json.dumps(agg))
