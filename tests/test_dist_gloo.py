"""The N>1 path on CPU: 2, 3 (ragged shards) and 8 ranks over the gloo backend gather their call tables, merge the sharded
select.reference.set and exchange the cohort's count slabs exactly as the RCCL path does on GPUs (same code,
exomedepth_amd/dist.py): the rank-offset arithmetic, shards of unequal width, ranks that own nothing."""
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


def _spawn(world, target, args, n_results):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(n_results)]
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return got


def _worker(rank, world, port, q, S_total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exomedepth_amd import api, dist as eddist
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


@pytest.mark.parametrize("world,S_total", [(2, 10), (3, 10), (8, 21), (8, 5)])     # (8, 5): three ranks own no sample at all
def test_gather_call_tables(world, S_total):
    got = _spawn(world, _worker, (S_total,), 1)[0]
    exp = []
    for s in range(S_total):
        for k in range(s % 3):
            exp.append((s, k % 2, 100 * s + k, 100 * s + k + 4, 1 + (k % 2), 5))
    assert got.tolist() == [list(r) for r in exp]     # ordered by global sample, sample ids shifted per rank


def test_shard_bounds_tile_the_axis():
    from exomedepth_amd import dist as eddist
    for n in (0, 1, 5, 8, 13, 1024, 8191):
        for w in (1, 2, 3, 8):
            b = [eddist.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _refset_table(R):
    """a synthetic table of raw per-prefix rows (what the GPU shares would produce)"""
    from exomedepth_amd import api
    rng = np.random.default_rng(11)
    rows = np.zeros(R, dtype=api.REFSET_DTYPE)
    rows["ref_index"] = rng.permutation(R)
    rows["correlation"] = np.sort(rng.uniform(0.9, 0.999, R))[::-1]
    rows["phi"] = rng.uniform(1e-3, 1e-2, R)
    rows["mean_p"] = 1.0 / (np.arange(R) + 2.0)            # falls below 0.05 at i = 19: the loop's early exit
    rows["median_depth"] = 100.0 * (np.arange(R) + 1)
    rows["ratio_sd"] = rng.uniform(1, 2, R)
    rows["expected_BF"] = 10 + 5 * np.sin(np.arange(R) / 3.0)
    return rows


def _refset_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exomedepth_amd import api, dist as eddist
    R = 31
    full = _refset_table(R)

    def compute_part(lo, hi):     # a rank only knows its own share of the statistics
        part = np.zeros(R, dtype=api.REFSET_DTYPE)
        for f in ("phi", "mean_p", "median_depth", "ratio_sd", "expected_BF"):
            part[f] = np.nan
            part[f][lo:hi] = full[f][lo:hi]
        part["ref_index"] = full["ref_index"]
        part["correlation"] = full["correlation"]
        return part

    res = eddist.select_reference_set_sharded(None, np.zeros((1, R), dtype=np.int32), compute_part=compute_part)
    q.put((rank, res["reference.choice"], res["summary.stats"].tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_select_reference_set_sharded_merge(world):
    from exomedepth_amd import api
    got = _spawn(world, _refset_worker, (), world)
    exp = api.refset_finalize(_refset_table(31))
    assert sorted(r for r, _, _ in got) == list(range(world))
    for rank, choice, raw in got:
        assert choice == exp["reference.choice"]
        assert raw == exp["summary.stats"].tobytes()         # every rank ends with the same, complete table
    st = exp["summary.stats"]
    assert np.isnan(st["expected_BF"][19:]).all() and not np.isnan(st["phi"][19]) and np.isnan(st["phi"][20:]).all()
    assert len(exp["reference.choice"]) == int(np.nanargmax(st["expected_BF"])) + 1


def _cohort_refsets_worker(rank, world, port, q, S, E):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exomedepth_amd import dist as eddist
    rng = np.random.default_rng(5)
    counts = rng.integers(0, 1000, (E, S)).astype(np.int32)       # every rank can rebuild the whole matrix: the check below uses it
    lo, hi = eddist.shard_bounds(S, rank, world)
    seen = {}

    def compute_range(all_counts, t0, t1):    # stands where the GPU entry runs: records what it was handed
        seen["all"] = all_counts.numpy().copy()
        seen["range"] = (t0, t1)
        ref = all_counts.sum(dim=1, keepdim=True) - all_counts[:, t0:t1]      # "all the others" as a stand-in aggregate reference
        return {"reference": ref, "n_chosen": np.full(t1 - t0, S - 1, dtype=np.int32)}

    res = eddist.cohort_reference_sets_sharded(torch.from_numpy(np.ascontiguousarray(counts[:, lo:hi])), S, compute_range=compute_range)
    ok = np.array_equal(seen["all"], counts) and seen["range"] == (lo, hi) and tuple(res["reference"].shape) == (E, hi - lo)
    ok = ok and np.array_equal(res["reference"].numpy(), counts.sum(axis=1, keepdims=True) - counts[:, lo:hi])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


# equal shards (one all_gather into the (S, E) buffer), ragged ones (padded to the widest), ranks without a column
@pytest.mark.parametrize("world,S", [(2, 12), (2, 13), (3, 12), (3, 13), (8, 16), (8, 21), (8, 5)])
def test_cohort_reference_sets_sharded_gathers_every_column(world, S):
    """the reference-set stage of a sample-sharded cohort (vignette/vignette.Rnw:390-402 needs ALL samples as candidates): every rank
    ends up with the whole (E, S) count matrix, column order intact, and is asked for exactly its own tests"""
    got = sorted(_spawn(world, _cohort_refsets_worker, (S, 37), world))
    assert got == [(r, True) for r in range(world)]
