"""TEST INFRASTRUCTURE: per-sample CPU work unit for bench.py's `cpu_baseline` leg (process-level sample
parallelism over the host cores; the reference itself is single-threaded, vignette/vignette.Rnw:390-431 loops over
samples).  Only bench.py's cpu_baseline leg imports this."""
import time

import numpy as np


def one_sample(args):
    """emissions + Viterbi + call table (+ Nelder-Mead stand-in fit) for one sample column; returns seconds."""
    test, ref, phi, p, chrom_off, start, end, fit = args
    from oracle import edoracle as eo
    t0 = time.perf_counter()
    if fit:
        eo.fit_nm(test, ref)
    ll, _ = eo.get_loglike_matrix(phi, p, test + ref, test, 1.0, eo.LIBM)
    eo.callcnvs(ll, chrom_off, start, end)
    return time.perf_counter() - t0


def all_cores(test_h, ref_h, p, phi, chrom_off, start, end, fit, cores):
    """Run `cores` samples (one per process) concurrently; returns (cells per second, wall seconds, processes)."""
    import concurrent.futures as cf
    import multiprocessing as mp

    n = min(cores, test_h.shape[1])
    jobs = [(np.ascontiguousarray(test_h[:, s]), np.ascontiguousarray(ref_h[:, s]), float(phi[s]), float(p[s]),
             chrom_off, start, end, bool(fit)) for s in range(n)]
    with cf.ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn")) as ex:
        list(ex.map(one_sample, jobs[:n]))          # start the workers, load the library (untimed)
        t0 = time.perf_counter()
        list(ex.map(one_sample, jobs))
        wall = time.perf_counter() - t0
    return test_h.shape[0] * n / wall, wall, n
