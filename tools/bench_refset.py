"""Timing of select.reference.set at BASELINE.json configs[4] scale on ONE GPU: 500 000 bins x 2048 candidate
references (synthetic).  Prints one JSON line.  (The 8-GPU decomposition of SURVEY.md 8e is not built yet.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import exomedepth_amd as ed

E = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reduced = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
lam = torch.empty(E, device=dev, dtype=torch.float32).log_normal_(float(np.log(60.0)), 0.7, generator=g)
sig = torch.linspace(0.02, 0.4, R, device=dev)[torch.randperm(R, device=dev, generator=g)]
test = torch.poisson(lam, generator=g).to(torch.int32)
refs = torch.empty((E, R), device=dev, dtype=torch.int32)
for lo in range(0, E, 16384):
    hi = min(lo + 16384, E)
    noise = torch.exp(torch.randn((hi - lo, R), device=dev, generator=g) * sig[None, :])
    refs[lo:hi] = torch.poisson(lam[lo:hi, None] * noise, generator=g).to(torch.int32)
torch.cuda.synchronize()
ed.select_reference_set(test[:20000].contiguous(), refs[:20000, :64].contiguous())   # warm-up
t0 = time.perf_counter()
out = ed.select_reference_set(test, refs, n_bins_reduced=reduced)
dt = time.perf_counter() - t0
st = out["summary.stats"]
print(json.dumps({"workload": "select.reference.set, %d bins x %d references, n.bins.reduced=%d" % (E, R, reduced),
                  "seconds": dt, "n_bins_selected": out["n.bins"], "n_chosen": len(out["reference.choice"]),
                  "bins*refs/s": out["n.bins"] * R / dt, "best_expected_BF": float(np.nanmax(st["expected_BF"])),
                  "n_nan_BF": int(np.isnan(st["expected_BF"]).sum())}))
