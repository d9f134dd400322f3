"""Concordance of the GPU path with the reference's arithmetic, whole batches, in the mode that is timed.

The device runs a batch (BASELINE geometry: 200 000 exons x N samples, 24 chromosomes) in emit mode `strict` (GSL's arithmetic
operation for operation with the portable log / exp) or `tables` (log-gamma difference tables, sample-major: what bench.py times);
EVERY column is then evaluated by the checker's LIBM flavour -- bit-identical to the reference's compiled special functions
(tests/test_oracle_ref.py) -- with the (phi, expected) the device used, and compared:
    log-likelihood values beyond 1e-10 RELATIVE (no absolute floor), values that would pass only through the old 1e-12 absolute
    floor, the largest relative difference, discordant Viterbi states, discordant call rows, and what the tables left to the strict
    arithmetic (ed_batch_table_stats).
--deep: the reference's workflow regime -- every sample's reference is the sum of 20 - 32 other samples of the cohort
(vignette/vignette.Rnw:390-402: deep references, expected ~0.03 - 0.05) instead of the synthetic reference matrix.
    python tools/concordance.py --samples 1024 --emit-mode tables [--deep] [--fit 1] > profiles/r05_concordance_....json"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--exons", type=int, default=200_000)
    ap.add_argument("--emit-mode", default="tables", choices=["strict", "tables", "tables-tile"])
    ap.add_argument("--fit", type=int, default=1, help="1: (phi, expected) fitted on the device (BASELINE configs[2]); 0: the generator's (configs[1])")
    ap.add_argument("--deep", action="store_true")
    ap.add_argument("--counts-bits", type=int, default=32, help="16: the device counts as uint16 (ed_batch_set_counts_bits; tables mode)")
    ap.add_argument("--depth", type=float, default=100.0, help="median reads per exon and test sample of the synthetic counts (from ~250 on the samples are tail samples: "
                    "Stirling's series beyond the LDS windows, DESIGN.md 4.13)")
    ap.add_argument("--seed", type=int, default=20250621)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--margins", action="store_true", help="also: for every decision ON the reference's Viterbi path (the state the path is in at an "
                    "observation choosing its predecessor, reference src/hmm.cpp:78-85) the gap between the winning and the runner-up candidate -- how far the "
                    "emission differences of the table mode are from moving a back-pointer (oracle/ed_oracle.c: edo_callcnvs_margins)")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import exomedepth_amd as ed
    from exomedepth_amd import synth
    from oracle import edoracle as eo
    from concurrent.futures import ThreadPoolExecutor

    S, E, C = args.samples, args.exons, 24
    chrom_off, start, end = synth.exon_design(E, C, args.seed)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, args.seed, mean_depth=args.depth)
    if args.deep:
        rng = np.random.default_rng(args.seed + 1)
        dt = torch.from_numpy(test).cuda()
        agg = torch.zeros_like(dt)
        for s in range(S):
            k = int(rng.integers(20, 33))
            others = rng.choice(np.delete(np.arange(S), s), size=min(k, S - 1), replace=False)
            agg[:, s] = dt[:, torch.from_numpy(others).cuda()].sum(dim=1)
        ref = agg.cpu().numpy().astype(np.int32)
        del dt, agg
        torch.cuda.empty_cache()
    mode = {"strict": 0, "tables-tile": 1, "tables": 2}[args.emit_mode]
    layout = 1 if mode == 2 else 0
    plan = ed.Plan(chrom_off, start, end)
    b = ed.Batch(plan, S)
    if mode:
        b.set_emit_mode(mode)
    b.set_counts_layout(layout)
    t_in, r_in = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
    if args.counts_bits == 16:
        assert layout == 1 and test.min() >= 0 and ref.min() >= 0 and max(test.max(), ref.max()) < 65536
        b.set_counts_bits(16)
        t_in, r_in = t_in.astype(np.uint16), r_in.astype(np.uint16)
    dt, dr = ed.DeviceArray(t_in), ed.DeviceArray(r_in)
    if args.fit or args.deep:
        dphi, dexp = ed.DeviceArray(np.zeros(S)), ed.DeviceArray(np.zeros(S))
        b.fit(dt, dr, dphi, dexp)
        b.run(dt, dr, dphi, dexp)
        phi, p = np.asarray(dphi.to_host()), np.asarray(dexp.to_host())
        n_unconv = int(b.fit_unconverged()[0])
    else:
        b.run(dt, dr, phi, p)
        n_unconv = 0
    tstats = b.table_stats() if mode else None
    n_tail = sum(1 for s in range(S) if b.table_windows(s)[3]) if mode == 2 else None
    n_tab = sum(1 for s in range(S) if b.table_dims(s)[0] > 0) if mode else None
    path, calls = b.path(), b.calls()
    ll = b.loglik()
    nerr = b.n_gsl_errors()
    b.close(); plan.close()
    order = np.argsort(calls["sample"], kind="stable")
    calls = calls[order]
    first = np.searchsorted(calls["sample"], np.arange(S + 1))
    t0 = time.time()

    def one(s):
        t, r = test[:, s], ref[:, s]
        ell, _ = eo.get_loglike_matrix(float(phi[s]), float(p[s]), t + r, t, 1.0, eo.LIBM)
        epath, ecalls = eo.callcnvs(ell, chrom_off, start, end)
        got = ll[:, :, s]
        d = np.abs(got - ell)
        same = (np.isnan(got) & np.isnan(ell)) | (got == ell)
        rel_ok = same | (d <= 1e-10 * np.abs(ell))
        floor_ok = same | (d <= np.maximum(1e-12, 1e-10 * np.abs(ell)))
        m = np.isfinite(ell) & (ell != 0)
        mx = float(np.max(d[m] / np.abs(ell[m]))) if m.any() else 0.0
        mine = calls[first[s]:first[s + 1]]
        g = {tuple(int(v) for v in row) for row in zip(mine["start_exon"] + 1, mine["end_exon"] + 1, mine["type"], mine["nexons"])}
        w = {tuple(int(v) for v in row[:4]) for row in ecalls}
        mg = eo.callcnvs_margins(ell, chrom_off, start, end) if args.margins else None
        return (int(np.sum(~rel_ok)), int(np.sum(floor_ok & ~rel_ok)), mx, int(np.sum(got.view(np.int64) == ell.view(np.int64))),
                int(np.sum(path[:, s].astype(np.int8) != epath)), len(g ^ w), len(w), float(np.min(np.abs(ell[m]))) if m.any() else 0.0, mg,
                float(np.max(d[m])) if m.any() else 0.0)

    nthr = args.threads or max(1, min(64, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)) - 1))
    with ThreadPoolExecutor(nthr) as ex:            # the checker is a C call: the GIL is released
        res = list(ex.map(one, range(S)))
    out = {"workload": "%d exons x %d samples, 24 chromosomes, %s; references: %s" % (
               E, S, "phi / expected fitted on the device (BASELINE configs[2])" if (args.fit or args.deep) else "phi / expected given (configs[1])",
               "sums of 20 - 32 other samples of the cohort (deep aggregate references, the reference's workflow)" if args.deep else "the synthetic reference matrix (K = 8)"),
           "emit_mode": args.emit_mode, "device_counts": "uint16" if args.counts_bits == 16 else "int32", "columns_compared": S,
           "compared_against": "checker, libm flavour (bit-identical to the reference's compiled lnbeta; C_hmm restated), given the (phi, expected) the device used",
           "cells": E * S, "loglik_values": 3 * E * S,
           "loglik_beyond_1e-10_relative": sum(r[0] for r in res),
           "loglik_passing_only_through_the_1e-12_absolute_floor": sum(r[1] for r in res),
           "max_relative_loglik_difference": max(r[2] for r in res),
           "smallest_nonzero_abs_loglik": min(r[7] for r in res),
           "loglik_bitwise_equal": sum(r[3] for r in res),
           "north_star_tolerance": "1e-10 relative",
           "discordant_viterbi_states": sum(r[4] for r in res), "columns_with_discordant_states": sum(1 for r in res if r[4]),
           "reference_call_rows": sum(r[6] for r in res), "discordant_call_rows": sum(r[5] for r in res),
           "n_gsl_errors": int(nerr), "fit_unconverged": n_unconv,
           "expected_range": [float(np.min(p)), float(np.max(p))], "phi_range": [float(np.min(phi)), float(np.max(phi))],
           "table_stats": tstats, "samples_on_tables": n_tab, "tail_samples": n_tail, "depth": args.depth,
           "cpu_threads": nthr, "cpu_seconds_wall": time.time() - t0}
    if args.margins:
        mgs = [r[8] for r in res]
        k = int(np.argmin([g["min_margin"] for g in mgs]))
        out["decision_margins"] = {
            "what": "every observation of every chain: the state the reference's Viterbi path is in chooses its predecessor among three candidates "
                    "(proba + vit[k]) + log(trans[k]), first strict maximum (reference src/hmm.cpp:78-85); margin = best - runner-up, on the checker's "
                    "libm-flavour matrix (= the reference's arithmetic)",
            "on_path_decisions": sum(g["decisions"] for g in mgs), "exact_ties": sum(g["ties"] for g in mgs),
            "nonzero_margins_below": {t: sum(g["below"][t] for g in mgs) for t in mgs[0]["below"]},
            "min_nonzero_margin": mgs[k]["min_margin"], "abs_score_at_that_decision": mgs[k]["scale_at_min"], "column_of_the_minimum": k,
            "max_abs_loglik_difference_device_vs_reference": max(r[9] for r in res),
            "reading": "a back-pointer can only differ between the device's table-mode matrix and the reference's if the candidates' difference moves by "
                       "more than the margin; the two matrices differ by at most max_abs_loglik_difference per emission, and two candidates of one decision "
                       "share the emission of the observation itself -- what separates them is the difference of two accumulated scores over the stretch "
                       "where their paths differ"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
