"""Randomised sweep of the table-driven emission modes against strict mode (mode 0 = the bit-level reference on the device) and,
on spot columns, against the LIBM flavour of the CPU checker (= the reference's arithmetic): odd shapes, sample counts around the tile
widths, empty chromosomes, depths from 2 to 3000 reads per exon, dispersions and proportions inside and outside what the tables
serve, tumour mixtures, negative counts, tiny table caps (most cells through the strict list / the full-scan fallback), both count
layouts, overlap groups on and off, repeated runs on one batch object.
    python tools/fuzz_tables.py [seconds] [seed]
Asserts: every log-likelihood within 1e-10 RELATIVE of strict mode's (NaN for NaN, equal values pass; no absolute floor -- the values
that would have needed the old 1e-12 floor are counted and must be 0); cells the tables do not serve (outside (Ly, Lr), under the
few-reads rule, or of samples without tables) bit-identical; the same GSL error counts; and reports the Viterbi states / call rows
that differ (expected: 0).  A third of the cases are the reference's workflow in small: references that are sums of 8 - 32 other
samples of the cohort (deep, expected 0.03 - 0.1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
from oracle import edoracle as eo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 424242)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def close(got, want):
    return (np.isnan(got) & np.isnan(want)) | (got == want) | (np.abs(got - want) <= 1e-10 * np.abs(want))


def close_floor(got, want):
    return (np.isnan(got) & np.isnan(want)) | (got == want) | (np.abs(got - want) <= np.maximum(1e-12, 1e-10 * np.abs(want)))


t0 = time.time()
n_cases = n_cells = n_disc_states = n_disc_calls = n_oracle_cols = n_cold = n_notab = n_samples = n_floor = n_u16 = 0
worst_a12 = 0.0
worst = 0.0
worst_at = None
while time.time() - t0 < budget:
    S = int(rng.choice([1, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 130, 257, 520]))
    E = int(rng.integers(1, 60 if S > 200 else 900) * rng.choice([1, 7]))
    C = int(rng.integers(1, 7))
    seed = int(rng.integers(1 << 30))
    chrom_off, start, end = synth.exon_design(max(E, C), C, seed)
    E = int(chrom_off[-1])
    if rng.random() < 0.3 and C > 1:
        k = int(rng.integers(1, C))
        chrom_off = np.insert(chrom_off, k, chrom_off[k]).astype(np.int32)
        C += 1
    depth = float(np.exp(rng.uniform(np.log(2.0), np.log(3000.0))))
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
    if rng.random() < 0.33 and S >= 9:                  # aggregate references: sums of other samples of the cohort
        agg = np.zeros_like(test)
        for j in range(S):
            k = int(rng.integers(8, min(33, S)))
            agg[:, j] = test[:, rng.choice(np.delete(np.arange(S), j), size=k, replace=False)].sum(axis=1)
            p[j] = float(test[:, j].sum() + 0.5) / float(test[:, j].sum() + agg[:, j].sum() + 1.0)
        ref = np.minimum(agg, 2**30).astype(np.int32)
    elif rng.random() < 0.3:
        test, ref, p = ref.copy(), test.copy(), 1.0 - p
    if rng.random() < 0.5:
        dead = rng.random(test.shape) < 0.2
        test[dead] = 0; ref[dead] = 0
    phi = np.minimum(phi * float(rng.choice([1.0, 1.0, 0.05, 1e-3, 30.0])), 0.7)
    if rng.random() < 0.3:                          # samples the tables do not serve, next to ones they do
        phi = phi.copy(); p = p.copy()
        for j in rng.choice(S, size=min(S, int(rng.integers(1, 5))), replace=False):
            m = int(rng.integers(0, 6))
            if m == 0: phi[j] = float(rng.uniform(1.0, 3.0))
            elif m == 1: p[j] = float(rng.choice([0.0, 1.0, 1.3, -0.1]))
            elif m == 2: phi[j] = float(rng.choice([0.0, 1.0, np.nan]))
            elif m == 3: p[j] = float(np.exp(rng.uniform(np.log(1e-8), np.log(1e-3))))
            elif m == 4: phi[j] = float(np.exp(rng.uniform(np.log(1e-9), np.log(1e-4))))
            else: p[j] = 1.0 - float(np.exp(rng.uniform(np.log(1e-8), np.log(1e-3))))
    if rng.random() < 0.2 and E > 3:
        test = test.copy(); test[int(rng.integers(E)), int(rng.integers(S))] = -int(rng.integers(1, 50))
    mixture = float(rng.choice([1.0, 1.0, 0.4]))
    caps = {}
    if rng.random() < 0.3:
        caps = dict(cap_obs=int(rng.choice([64, 128, 4096])), cap_ref=int(rng.choice([64, 1024, 32768])), reach=float(rng.choice([1.0, 2.0, 8.0, 30.0])))
    plan = ed.Plan(chrom_off, start, end, float(rng.choice([1e-4, 1e-2])), float(rng.choice([5e4, 2e3])))
    tp, L = plan.transition_probability, plan.expected_CNV_length
    overlap = int(rng.integers(0, 2))
    ref_b = ed.Batch(plan, S); ref_b.set_viterbi_overlap(overlap)
    ref_b.run(test, ref, phi, p, mixture=mixture)
    ll0, path0, calls0, nerr0 = ref_b.loglik(), ref_b.path(), ref_b.calls(), ref_b.n_gsl_errors()
    ref_b.close()
    for mode in ((2, 1) if rng.random() < 0.35 else (2,)):
        layout = int(rng.integers(0, 2)) if mode == 2 else 0
        b = ed.Batch(plan, S); b.set_viterbi_overlap(overlap)
        b.set_emit_mode(mode, **caps)
        b.set_counts_layout(layout)
        t_in, r_in = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
        if layout == 1 and rng.random() < 0.5 and test.min() >= 0 and ref.min() >= 0 and max(test.max(), ref.max()) < 65536:
            b.set_counts_bits(16)                   # the 16-bit device format: the same bits as int32
            t_in, r_in = t_in.astype(np.uint16), r_in.astype(np.uint16)
            n_u16 += 1
        for _ in range(int(rng.integers(1, 3))):
            b.run(t_in, r_in, phi, p, mixture=mixture)
        ll, path, calls = b.loglik(), b.path(), b.calls()
        ok = close(ll, ll0)
        if not ok.all():
            bad = np.argwhere(~ok)
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez_compressed("gpurun_out/fuzz_tables_case.npz", chrom_off=chrom_off, start=start, end=end, test=test, ref=ref, p=p, phi=phi, mixture=mixture)
            for e, st, s in bad[:8]:
                print("  exon", e, "state", st, "sample", s, "obs", test[e, s], "ref", ref[e, s], "phi", phi[s], "p", p[s], "got %r" % ll[e, st, s], "strict %r" % ll0[e, st, s])
            raise AssertionError(("loglik", mode, layout, E, S, C, seed, len(bad)))
        fin = np.isfinite(ll0) & (ll0 != 0)
        if fin.any():
            rel = np.where(fin, np.abs(ll - ll0) / np.where(fin, np.abs(ll0), 1.0), 0.0)
            w = float(rel.max())
            if w > worst:
                worst = w
                e_, st_, s_ = (int(v) for v in np.unravel_index(int(np.argmax(rel)), rel.shape))
                worst_at = dict(mode=mode, phi=float(phi[s_]), p=float(p[s_]), state=st_, obs=int(test[e_, s_]), ref=int(ref[e_, s_]), strict=float(ll0[e_, st_, s_]),
                                mixture=mixture, a12=(1.0 - float(phi[s_])) / float(phi[s_]))
        n_floor += int(np.sum(close_floor(ll, ll0) & ~ok))
        st = b.table_stats()
        n_cold += st["n_cold_cells"]; n_notab += st["n_samples_without_tables"]; n_samples += S
        for s in range(S):                          # what the tables do not serve: strict bits
            ly, lr, tm1, w = b.table_dims(s)
            t64, r64 = test[:, s].astype(np.int64), ref[:, s].astype(np.int64)
            out = ~((t64 >= 0) & (t64 < ly) & (r64 >= 0) & (r64 < lr)) | ((t64 + r64 >= 1) & (t64 + r64 <= tm1)) | ((r64 == 0) & (t64 >= w))
            assert np.array_equal(bits(ll[out, :, s]), bits(ll0[out, :, s])), ("strict bits", mode, E, S, seed, s, ly, lr, tm1, w)
        assert b.n_gsl_errors() == nerr0, ("nerr", mode, E, S, seed)
        d = int(np.sum(path != path0))
        n_disc_states += d
        n_disc_calls += len({tuple(int(v) for v in r) for r in calls} ^ {tuple(int(v) for v in r) for r in calls0})
        if d:
            print("  discordant states:", d, "case", (mode, layout, E, S, C, seed))
        for s in rng.choice(S, size=min(S, 2), replace=False):      # spot columns against the reference's arithmetic
            if b.emit_tables(int(s))[0] == 0:
                continue
            ell, _ = eo.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], mixture, eo.LIBM)
            okl = close(ll[:, :, s], ell)
            if not okl.all():
                for e, st_ in np.argwhere(~okl)[:8]:
                    print("  libm: exon", e, "state", st_, "sample", s, "obs", test[e, s], "ref", ref[e, s], "phi", phi[s], "p", p[s], "got %r" % ll[e, st_, s], "libm %r" % ell[e, st_])
            assert okl.all(), ("libm", mode, E, S, seed, int(s), phi[s], p[s])
            n_oracle_cols += 1
        b.close()
        n_cells += E * S
    plan.close()
    n_cases += 1
print("fuzz_tables ok: %d cases, %d cells in table modes, max relative difference from strict mode %.2e (bar 1e-10, no absolute floor; %d values would have "
      "needed the 1e-12 floor), %d discordant Viterbi states, %d discordant call rows, %d columns against the checker's libm flavour, "
      "%d cells on the strict lists, %d of %d samples without tables, %d runs on 16-bit counts, %.0f s"
      % (n_cases, n_cells, worst, n_floor, n_disc_states, n_disc_calls, n_oracle_cols, n_cold, n_notab, n_samples, n_u16, time.time() - t0))
print("  the largest difference:", worst_at)
