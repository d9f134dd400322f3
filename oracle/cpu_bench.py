"""TEST INFRASTRUCTURE: all-core leg of bench.py's `cpu_baseline` (process-level sample parallelism over the host cores;
the reference itself is single-threaded, vignette/vignette.Rnw:390-431 loops over samples).  Only bench.py imports this.

One worker process per core, pinned to it (sched_setaffinity); every worker loads the checker, waits at a barrier, then
times ITS OWN loop over the samples it was dealt -- process start-up, pickling and library loading are outside the
timed region.  Rate = cells of all workers / the slowest worker's time."""
import os
import time

import numpy as np


def _worker(core, barrier, queue, cols, chrom_off, start, end, fit):
    try:
        os.sched_setaffinity(0, {core})
    except (AttributeError, OSError):
        pass
    from oracle import edoracle as eo
    eo.lib()
    barrier.wait()
    t0 = time.perf_counter()
    for test, ref, phi, p in cols:
        if fit:
            eo.fit_nm(test, ref)
        ll, _ = eo.get_loglike_matrix(phi, p, test + ref, test, 1.0, eo.LIBM)
        eo.callcnvs(ll, chrom_off, start, end)
    queue.put((core, time.perf_counter() - t0, len(cols)))


def cpu_quota():
    """CPU bandwidth the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None if unlimited or unknown.
    A box can show 256 logical CPUs in its affinity mask and still be throttled to a dozen cores' worth of time."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def all_cores(test_h, ref_h, p, phi, chrom_off, start, end, fit, max_workers):
    import math
    import multiprocessing as mp

    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    n_affinity = len(cores)
    quota = cpu_quota()
    if quota is not None:
        max_workers = min(max_workers, max(1, int(math.ceil(quota))))     # more workers than the quota only measure throttling
    cores = cores[:max_workers]
    n_s = test_h.shape[1]
    ctx = mp.get_context("spawn")
    barrier = ctx.Barrier(len(cores))
    queue = ctx.Queue()
    procs = []
    for i, core in enumerate(cores):
        s = i % n_s      # one sample per worker (same work per worker; samples are reused when there are more cores)
        cols = [(np.ascontiguousarray(test_h[:, s]), np.ascontiguousarray(ref_h[:, s]), float(phi[s]), float(p[s]))]
        pr = ctx.Process(target=_worker, args=(core, barrier, queue, cols, chrom_off, start, end, bool(fit)))
        pr.start()
        procs.append(pr)
    res = [queue.get() for _ in procs]
    for pr in procs:
        pr.join()
    slowest = max(r[1] for r in res)
    n_done = sum(r[2] for r in res)
    return {"value": test_h.shape[0] * n_done / slowest, "unit": "exons*samples/s", "cores": len(cores),
            "logical_cpus_in_affinity_mask": n_affinity, "cgroup_cpu_quota_cores": quota,
            "slowest_worker_s": slowest, "mean_worker_s": float(np.mean([r[1] for r in res])),
            "sample": "%d workers pinned one per logical core, one sample column of %d exons each, timed inside the workers "
                      "after a common barrier" % (len(cores), test_h.shape[0])}
