"""Timing of select.reference.set at BASELINE.json configs[4] scale: 500 000 bins x 2048 candidate references
(synthetic).  Prints one JSON line (rank 0).
  python tools/bench_refset.py [E R n_bins_reduced]                                  one GPU
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_refset.py ...
        N ranks: every rank holds the matrix, fits a share of the sorted prefixes, one all_gather of the rows
        (exomedepth_amd/dist.py::select_reference_set_sharded).  ED_BENCH_BACKEND=gloo ED_BENCH_SHARE_GPU=1 run the
        same code on a 1-GPU box (functional check only)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import exomedepth_amd as ed

E = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reduced = int(sys.argv[3]) if len(sys.argv) > 3 else 0
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = 0 if os.environ.get("ED_BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
if world > 1:
    import torch.distributed as dist
    from exomedepth_amd import dist as eddist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(os.environ.get("ED_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
dev = torch.device("cuda", local_rank)
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
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
if world > 1:
    out = eddist.select_reference_set_sharded(test, refs, n_bins_reduced=reduced)
    out["n.bins"] = -1
    dist.barrier()
else:
    out = ed.select_reference_set(test, refs, n_bins_reduced=reduced)
dt = time.perf_counter() - t0
st = out["summary.stats"]
if rank == 0:
  print(json.dumps({"workload": "select.reference.set, %d bins x %d references, n.bins.reduced=%d" % (E, R, reduced),
                  "seconds": dt, "n_bins_selected": out["n.bins"], "n_chosen": len(out["reference.choice"]),
                  "n_gpus": world, "bins*refs/s": E * R / dt, "best_expected_BF": float(np.nanmax(st["expected_BF"])),
                  "n_nan_BF": int(np.isnan(st["expected_BF"]).sum())}))
if world > 1:
    dist.destroy_process_group()
