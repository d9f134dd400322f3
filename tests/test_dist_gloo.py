"""The N>1 path on CPU: two ranks over the gloo backend gather their call tables exactly as the RCCL
path does on GPUs (same code, exomedepth_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exomedepth_amd import api, dist as eddist
    S_total = 10
    lo, hi = eddist.shard_bounds(S_total, rank, world)
    # each rank fabricates the call table its shard would produce: sample s has (s % 3) calls
    rows = []
    for s in range(lo, hi):
        for k in range(s % 3):
            rows.append((s - lo, k % 2, 100 * s + k, 100 * s + k + 4, 1 + (k % 2), 5))
    calls = np.array(rows, dtype=api.CALL_DTYPE) if rows else np.zeros(0, dtype=api.CALL_DTYPE)
    t = eddist.calls_to_tensor(calls, torch.device("cpu"))
    g = eddist.gather_call_tables(t, lo)
    if rank == 0:
        q.put(g.numpy())
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_call_tables_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    exp = []
    for s in range(10):
        for k in range(s % 3):
            exp.append((s, k % 2, 100 * s + k, 100 * s + k + 4, 1 + (k % 2), 5))
    assert got.tolist() == [list(r) for r in exp]     # ordered by global sample, sample ids shifted per rank
