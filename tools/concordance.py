"""Concordance of the GPU path with the reference's arithmetic: the device computes with the
portable log/exp (bit-identical to the checker's portable flavour); the reference calls libm.  This tool runs the
GPU on BASELINE configs[1] geometry (200 000 exons x 64 samples, given phi) and the checker's LIBM flavour -- which is
bit-identical to the reference's compiled special functions (tests/test_oracle_ref.py) -- on the same columns, and
reports: cells compared, max relative log-likelihood difference, discordant Viterbi states, discordant call rows.
    python tools/concordance.py [n_samples] > gpurun_out/concordance.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one(args):
    test, ref, phi, p, chrom_off, start, end = args
    from oracle import edoracle as eo
    ll, _ = eo.get_loglike_matrix(phi, p, test + ref, test, 1.0, eo.LIBM)
    path, calls = eo.callcnvs(ll, chrom_off, start, end)
    return ll, path, calls


if __name__ == "__main__":
    import torch
    torch.cuda.init()
    import exomedepth_amd as ed
    from exomedepth_amd import synth
    import concurrent.futures as cf
    import multiprocessing as mp

    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    E, C = 200_000, 24
    chrom_off, start, end = synth.exon_design(E, C, 20250621)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 20250621)
    plan = ed.Plan(chrom_off, start, end)
    batch = ed.Batch(plan, S)
    batch.run(test, ref, phi, p)
    ll, path, calls = batch.loglik(), batch.path(), batch.calls()
    batch.close(); plan.close()
    t0 = time.time()
    jobs = [(np.ascontiguousarray(test[:, s]), np.ascontiguousarray(ref[:, s]), float(phi[s]), float(p[s]), chrom_off, start, end)
            for s in range(S)]
    with cf.ProcessPoolExecutor(max_workers=min(S, os.cpu_count() or 1), mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(one, jobs))
    max_rel = 0.0
    bad_states = bad_calls = n_calls_ref = bitwise_equal = 0
    for s, (ell, epath, ecalls) in enumerate(res):
        g = ll[:, :, s]
        d = np.abs(g - ell)
        den = np.maximum(np.abs(ell), 1e-300)
        m = np.isfinite(ell) & (ell != 0)
        max_rel = max(max_rel, float(np.max(d[m] / den[m])) if m.any() else 0.0)
        bitwise_equal += int(np.sum(g.view(np.int64) == ell.view(np.int64)))
        bad_states += int(np.sum(path[:, s].astype(np.int8) != epath))
        mine = calls[calls["sample"] == s]
        n_calls_ref += len(ecalls)
        same = len(mine) == len(ecalls) and np.array_equal(mine["start_exon"] + 1, ecalls[:, 0].astype(np.int64)) and \
            np.array_equal(mine["end_exon"] + 1, ecalls[:, 1].astype(np.int64)) and np.array_equal(mine["type"], ecalls[:, 2].astype(np.int64))
        if not same:
            bad_calls += abs(len(mine) - len(ecalls)) + (int(np.sum(mine["start_exon"][: min(len(mine), len(ecalls))] + 1 != ecalls[: min(len(mine), len(ecalls)), 0])) if len(mine) and len(ecalls) else 0)
    print(json.dumps({"workload": "200000 exons x %d samples, 24 chromosomes, phi given (BASELINE configs[1] geometry)" % S,
                      "compared_against": "checker, libm flavour (bit-identical to the reference's compiled lnbeta; C_hmm restated)",
                      "cells": E * S, "loglik_values": 3 * E * S, "loglik_bitwise_equal": bitwise_equal,
                      "max_relative_loglik_difference": max_rel, "north_star_tolerance": 1e-10,
                      "discordant_viterbi_states": bad_states, "reference_call_rows": n_calls_ref,
                      "discordant_call_rows": bad_calls, "cpu_seconds_wall": time.time() - t0}))
