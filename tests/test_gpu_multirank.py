"""The N > 1 code path of bench.py on a 1-GPU box: two ranks launched exactly as the driver launches them
(python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2), both on GPU 0 (ED_BENCH_SHARE_GPU=1), the gather of the
call tables -- the path's only collective -- over gloo (RCCL refuses two ranks on one device).  Samples are sharded by rank with no
data-path collective; the gathered table must hold exactly the calls of both shards."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_one_gpu_and_gather_their_call_tables(edlib):
    torch = pytest.importorskip("torch")
    from exomedepth_amd import synth
    E, S, C = 20000, 128, 24
    env = dict(os.environ, ED_BENCH_SHARE_GPU="1", ED_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--exons", str(E), "--samples", str(S), "--cpu-samples", "0", "--verify-columns", "0", "--fit-concordance", "0",
           "--config1-steps", "0", "--kernel-alone", "0", "--stage-inputs", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["samples_total"] == 2 * S
    assert d["value"] > 0 and abs(d["value"] - E * S * 2 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    # what the two shards produce, one after the other in this process (the seeds bench.py gives ranks 0 and 1)
    dev = torch.device("cuda", 0)
    chrom_off, start, end = synth.exon_design(E, C, seed=20250620)
    plan = edlib.Plan(chrom_off, start, end, 1e-4, 50000.0)
    want = 0
    tests = []
    for rank in (0, 1):
        torch.manual_seed(20250620 + 3 + rank)      # (bench.py seeds the global generator too: the gamma variates draw from it)
        test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=20250620 + 3 + 1000 * rank, mean_depth=100.0)
        tests.append(test)
        b = edlib.Batch(plan, S)
        b.set_emit_mode(2)       # (bench.py's default mode; the fit -- and with it the calls -- is the same in every mode to 1e-9)
        dphi = torch.empty(S, dtype=torch.float64, device=dev); dexp = torch.empty(S, dtype=torch.float64, device=dev)
        b.fit(test, ref, dphi, dexp)
        b.run(test, ref, dphi, dexp)
        want += b.n_calls()
        b.close()
    plan.close()
    assert d["n_calls"] == want and want > 0
    # the workflow leg at N = 2: the reference-set stage sharded by tests (every rank: its own columns against ALL 2 S candidates, one
    # all_gather of the count slabs) must choose, for rank 0's samples, what ONE call on the whole cohort chooses for them
    w = d["extra"]["workflow"]
    assert w is not None and w["ranks"] == 2 and w["n_calls"] > 0
    whole = torch.cat(tests, dim=1).contiguous()
    bl = (np.asarray(end) - np.asarray(start)) / 1000.0
    rs = edlib.cohort_select_reference_sets(whole, bl, 10000, max_refs=32, want_reference=False)
    ch = rs["choice"][:S]
    assert w["choice_checksum_rank0"] == int(np.sum((ch.astype(np.int64) + 1) * (np.arange(ch.shape[1], dtype=np.int64) + 1)[None, :]))
    assert abs(w["references_chosen_mean"] - float(rs["n_chosen"][:S].mean())) < 1e-12


def test_one_rank_over_rccl(edlib):
    """A process group of ONE rank over RCCL on the box's GPU (ED_BENCH_FORCE_PG=1): the collectives of the N > 1 path as RCCL runs them -- barrier, the
    MAX all-reduce of the clock, the gather of the call tables from DEVICE memory, and the workflow leg's all_gather_into_tensor of the count slabs -- the
    branches the two-ranks-on-one-GPU test (gloo, host memory) cannot reach.  Same calls, same reference choices as the run without a group."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--exons", "20000", "--samples", "128",
            "--cpu-samples", "0", "--verify-columns", "0", "--fit-concordance", "0", "--config1-steps", "0", "--kernel-alone", "0", "--stage-inputs", "0",
            "--strict-steps", "0", "--workflow-reps", "1"]
    out = {}
    for name, extra in (("plain", {}), ("rccl", {"ED_BENCH_FORCE_PG": "1", "MASTER_PORT": "29647"})):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        r = subprocess.run(base, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # ONE line on stdout, RCCL's banner (NCCL_DEBUG=VERSION on the boxes) included nowhere
        out[name] = json.loads(lines[0])
    a, b = out["plain"], out["rccl"]
    assert b["n_gpus"] == 1 and b["n_calls"] == a["n_calls"] > 0
    wa, wb = a["extra"]["workflow"], b["extra"]["workflow"]
    assert wa["sharding"] is None and wb["sharding"] is not None            # the group's run went through dist.cohort_reference_sets_sharded
    assert wb["choice_checksum_rank0"] == wa["choice_checksum_rank0"] and wb["n_calls"] == wa["n_calls"] > 0
    assert abs(wb["references_chosen_mean"] - wa["references_chosen_mean"]) < 1e-12
