"""Multi-GPU layout of the path: samples are independent (reference vignette loop,
vignette/vignette.Rnw:390-431), so each rank owns a contiguous slab of sample columns and runs the
whole pipeline locally; the exon design (plan) is replicated.  The only collective is the final
gather of the compact call tables (KBs per rank) -- paths and likelihoods stay resident on their GPU.

One process per GPU; torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm).  The same
code runs on the "gloo" backend with CPU tensors, which is how tests/test_dist_gloo.py covers it.
"""
import numpy as np


def shard_bounds(n_samples, rank, world_size):
    """Contiguous, balanced slab [lo, hi) of sample columns owned by `rank`."""
    base, rem = divmod(int(n_samples), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_call_tables(local_calls, sample_offset, group=None, dst=0):
    """Gather per-rank call tables on `dst` -- the path's only collective (RCCL over xGMI: KBs per rank).

    local_calls: torch int32 tensor [n_local, 6] (sample, chrom, start_exon, end_exon, type, nexons) on the backend's
    device (device_call_table() gives it straight from a batch's device-resident table); `sample` is local to the rank
    and is shifted by sample_offset so that the gathered table indexes the global sample axis.  Returns the
    concatenated [n_total, 6] tensor on `dst` (ordered by rank, hence by global sample) and None elsewhere.
    One `gather` of the row counts to `dst`, then every other rank sends exactly its rows to `dst` (point to point:
    xGMI is a full mesh, each sender has its own link) -- nothing is padded and nothing goes to ranks that do not
    need it."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)        # group-local; `dst` is group-local too
    # gather / send / recv address GLOBAL ranks: translate the group-local ones (identity for the default group)
    glob = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    dev = local_calls.device
    rows = local_calls.clone()
    if rows.numel():
        rows[:, 0] += int(sample_offset)
    n_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(n_local, counts, dst=glob(dst), group=group)
    if rank != dst:
        if rows.shape[0]:
            dist.send(rows.contiguous(), dst=glob(dst), group=group)
        return None
    parts = []
    for r in range(world):
        c = int(counts[r].item())
        if r == dst:
            parts.append(rows)
        elif c:
            buf = torch.empty((c, 6), dtype=torch.int32, device=dev)
            dist.recv(buf, src=glob(r), group=group)
            parts.append(buf)
    return torch.cat(parts, dim=0) if parts else rows


class _DevicePointer:
    """a raw device buffer as a __cuda_array_interface__ object (so that torch can wrap it without a copy through the host)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_call_table(batch):
    """The call table of the batch's last run as an int32 [n, 6] torch CUDA tensor, copied device-to-device from the
    library's table (no round trip through the host).  Synchronises the batch (the row count is host data)."""
    import torch

    n = batch.n_calls()
    if n == 0:
        return torch.zeros((0, 6), dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    ptr = batch.device_pointers()["calls"]
    view = torch.as_tensor(_DevicePointer(ptr, (n, 6), "<i4"), device=torch.device("cuda", torch.cuda.current_device()))
    return view.clone()


def calls_to_tensor(calls_np, device):
    """Structured numpy call table (api.CALL_DTYPE) -> int32 [n,6] torch tensor on `device`."""
    import torch

    a = np.ascontiguousarray(calls_np).view(np.int32).reshape(-1, 6)
    return torch.from_numpy(a.copy()).to(device)


def select_reference_set_sharded(test_counts, reference_counts, bin_length=None, n_bins_reduced=0, names=None, group=None,
                                 compute_part=None):
    """select.reference.set over the ranks of `group` (BASELINE configs[4]: 500 000 bins x 2 048 references).

    Every rank holds the (E, R) count matrix (4 GB at that size: replicated, not sharded, in 288 GB of HBM) and
    fits a contiguous share of the R cumulative references of the correlation-sorted axis -- the prefixes are
    independent given the order -- so the only exchange is one all_gather of the per-prefix result rows
    (R x 56 bytes).  The loop's early exit and the choice are applied to the merged table on every rank.
    compute_part(begin, end) -> REFSET_DTYPE rows: defaults to the GPU path; injectable so that the merge logic is
    testable on the gloo backend without a GPU."""
    import torch
    import torch.distributed as dist
    from . import api

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    R = int(reference_counts.shape[1])
    lo, hi = shard_bounds(R, rank, world)
    low_cov = False
    if hi > lo:
        if compute_part is None:
            part = api.select_reference_set(test_counts, reference_counts, bin_length, n_bins_reduced, names, prefix_window=(lo, hi))
            rows, low_cov = part["summary.stats"], part["low.coverage"]
        else:
            rows = compute_part(lo, hi)
    else:
        rows = np.zeros(R, dtype=api.REFSET_DTYPE)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    nf = api.REFSET_DTYPE.itemsize // 8
    mine = torch.from_numpy(np.ascontiguousarray(rows).view(np.float64).reshape(R, nf).copy()).to(dev)
    bufs = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    merged = np.zeros(R, dtype=api.REFSET_DTYPE)
    flat = merged.view(np.float64).reshape(R, nf)
    src = None
    for r in range(world):
        a, b = shard_bounds(R, r, world)
        if b > a:
            flat[a:b] = bufs[r][a:b].cpu().numpy()
            src = r if src is None else src
    # ref_index / correlation are the same on every rank that computed anything: take them from the first one
    first = bufs[src].cpu().numpy().view(api.REFSET_DTYPE).reshape(R)
    merged["ref_index"] = first["ref_index"]
    merged["correlation"] = first["correlation"]
    if low_cov:
        nm = names if names is not None else ["X%d" % (i + 1) for i in range(R)]
        return {"reference.choice": [nm[int(merged["ref_index"][0])]], "summary.stats": merged}
    return api.refset_finalize(merged, names)


def cohort_reference_sets_sharded(local_counts, n_samples_total, bin_length=None, n_bins_reduced=0, max_refs=32, group=None,
                                  compute_range=None):
    """The reference-set stage of the workflow for a SAMPLE-SHARDED cohort (BASELINE configs[3]: 8192 samples over 8 GPUs).

    In the reference every test sample runs select.reference.set against ALL the other samples (vignette/vignette.Rnw:390-402,
    R/optimize_reference_set.R:100-141), so a rank that owns the columns shard_bounds(S, rank, world) needs the candidates that live
    on the other ranks.  One `all_gather` of the count slabs gives every rank the whole (E, S) matrix -- 6.5 GB at 200 000 x 8192
    int32, i.e. each GPU takes in 7/8 of it over its 7 xGMI links (~5.7 GB at a few hundred GB/s: tens of milliseconds, once per
    cohort; the slabs are (E, S_r) column blocks, so they are gathered as their transposes, contiguous (S_r, E) blocks of the
    (S, E) matrix, and the result is viewed back) -- and the rank then runs ed_cohort_select_reference_sets_range on ITS tests:
    a (S_r x S) block of the correlation matrix, its tests' prefixes / fits, and the aggregate references of its own columns,
    which is exactly what its ed_cohort_submit needs next to its own counts.  No other exchange.

    local_counts: (E, S_r) int32 tensor on the backend's device.  Returns dict(n_chosen (S_r,), choice (S_r, K) with columns of the
    WHOLE cohort, summary.stats, n.bins, reference: (E, S_r) aggregate references).  compute_range(all_counts, t0, t1) -> that dict:
    defaults to the GPU path; injectable so that the gather / view logic is testable on the gloo backend without a GPU."""
    import torch
    import torch.distributed as dist
    from . import api

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    S = int(n_samples_total)
    E = int(local_counts.shape[0])
    t0, t1 = shard_bounds(S, rank, world)
    assert int(local_counts.shape[1]) == t1 - t0, "local_counts must hold this rank's columns shard_bounds(S, rank, world)"
    # gather the TRANSPOSED slabs: rank r's (S_r, E) block is rows [lo_r, hi_r) of the (S, E) matrix -- contiguous pieces of one buffer
    home = local_counts.device
    # (the gloo backend moves host memory: the slabs go through the host then -- what the 2-ranks-on-one-GPU test and CPU tests do)
    cdev = home if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = local_counts.t().contiguous().to(cdev)
    full_t = torch.empty((S, E), dtype=local_counts.dtype, device=cdev)
    pieces = [full_t[slice(*shard_bounds(S, r, world))] for r in range(world)]
    if all(p.shape == pieces[0].shape for p in pieces):
        dist.all_gather_into_tensor(full_t, mine, group=group) if hasattr(dist, "all_gather_into_tensor") and dist.get_backend(group) == "nccl" \
            else dist.all_gather(pieces, mine, group=group)
    else:                                   # ragged shards: all_gather wants equal shapes -- pad to the widest
        width = max(p.shape[0] for p in pieces)
        pad = torch.zeros((width, E), dtype=mine.dtype, device=mine.device)
        pad[:mine.shape[0]] = mine
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        for r in range(world):
            pieces[r].copy_(bufs[r][:pieces[r].shape[0]])
    all_counts = full_t.to(home).t().contiguous()    # (E, S): the layout the reference-set entry takes
    del full_t
    if compute_range is None:
        return api.cohort_select_reference_sets(all_counts, bin_length, n_bins_reduced, max_refs, want_reference=True, test_range=(t0, t1))
    return compute_range(all_counts, t0, t1)
