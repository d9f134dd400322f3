"""Table-driven emission mode (ed_batch_set_emit_mode(batch, 1); csrc/edtab.inc, csrc/ed_dtab.h) against the checker.

Bars (BASELINE.json north_star): log-likelihoods within 1e-10 relative of the reference's arithmetic -- here the LIBM flavour of
the checker, which is bit-identical to the reference's compiled special functions (tests/test_oracle_ref.py) -- and Viterbi paths
/ call tables identical.  The strict mode (mode 0) stays the bit-level reference; cells the tables do not serve must carry its bits.
"""
import numpy as np
import pytest

import exomedepth_amd as ed
from exomedepth_amd import synth

pytestmark = pytest.mark.gpu

REL_TOL = 1e-10   # north_star tolerance on log-likelihoods
ABS_TOL = 0.0     # no absolute floor: the bar is relative (equal values -- exact zeros, infinities -- and NaN for NaN pass)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def close(got, want):
    both_nan = np.isnan(got) & np.isnan(want)
    ok = both_nan | (got == want) | (np.abs(got - want) <= np.maximum(ABS_TOL, REL_TOL * np.abs(want)))
    return ok


def close_rel(got, want):
    """north_star's bar as written: 1e-10 RELATIVE, no absolute floor (equal values -- zeros, infinities -- and NaN for NaN pass)"""
    return (np.isnan(got) & np.isnan(want)) | (got == want) | (np.abs(got - want) <= REL_TOL * np.abs(want))


MODES = (1, 2)     # 1: exon-major tiles, tables through the caches; 2: sample-major, tables in LDS


def not_served(b, test, ref):
    """[E][S] mask of the cells the tables of the last run do not serve (ed_batch_copy_table_dims: outside (Ly, Lr), 0 < tot <= Tm1, or ref = 0 with obs >= N0),
    and the [S] mask of samples without tables"""
    E, S = test.shape
    out = np.zeros((E, S), dtype=bool)
    notab = np.zeros(S, dtype=bool)
    for s in range(S):
        ly, lr, tm1, w = b.table_dims(s)
        t, r = test[:, s].astype(np.int64), ref[:, s].astype(np.int64)
        out[:, s] = ~((t >= 0) & (t < ly) & (r >= 0) & (r < lr)) | ((t + r >= 1) & (t + r <= tm1)) | ((r == 0) & (t >= w))
        notab[s] = ly == 0
        assert w in (1, 2, 3, 4) if ly == 0 else w >= 1
    return out, notab


def run_modes(plan, S, test, ref, phi, p, mode=1, **tab_opts):
    """results of strict mode under key 0 and of table mode `mode` under key 1"""
    out = {}
    for key, m in ((0, 0), (1, mode)):
        b = ed.Batch(plan, S)
        if m:
            b.set_emit_mode(m, **tab_opts)
        b.run(test, ref, phi, p)
        out[key] = dict(ll=b.loglik(), path=b.path(), calls=b.calls(), nerr=b.n_gsl_errors(), batch=b)
    return out


@pytest.mark.parametrize("mode", MODES)
def test_tables_equal_the_host_definition(edlib, oracle, mode):
    """k_tab_build (parallel double-double scan) holds what csrc/ed_dtab.h defines (sequential, compiled by gcc into the checker)."""
    E, S = 3000, 40
    chrom_off, start, end = synth.exon_design(E, 3, 5)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 5, n_segments=2, mean_depth=150.0)
    plan = ed.Plan(chrom_off, start, end)
    b = ed.Batch(plan, S)
    b.set_emit_mode(mode)
    b.run(test, ref, phi, p)
    n_diff = n_all = 0
    for s in (0, 7, S - 1):
        ly, lr, t1, t2, t3 = b.emit_tables(s)
        assert ly >= 64 and lr >= 64 and ly % 8 == 0 and lr % 8 == 0
        assert t3.shape[0] == ly + lr
        e = p[s]
        sd2 = phi[s] * e * (1 - e)
        for st, odds in enumerate((0.5, 1.0, 1.5)):
            ep = e if st == 1 else (e * odds) / ((e * odds + 1) - e)
            a1 = ((ep * ep) * (1 - ep)) / sd2 - ep        # src/CNV_estimate.cpp:45-46 (sd * sd vs sd2: the device takes sqrt then squares)
            # the device's a1 / a2 come from k_sample_consts; rebuild them the same way to the last bit
            sd = np.sqrt((phi[s] * e) * (1.0 - e))
            a1 = ((ep * ep) * (1 - ep)) / (sd * sd) - ep
            a2 = ((1 - ep) / ep) * a1
            for tab, x0 in ((t1, a1), (t2, a2), (t3, a1 + a2)):
                want = oracle.dtab(x0, tab.shape[0])
                d = bits(tab[:, st]) != bits(want)
                n_diff += int(d.sum()); n_all += d.size
                # a different association of double-double additions may move an entry that sits on a rounding boundary by one ulp
                assert np.all(np.abs(tab[:, st] - want) <= np.spacing(np.abs(want))), (s, st)
    assert n_diff <= n_all // 1000, (n_diff, n_all)
    b.close(); plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_tables_mode_against_the_reference_arithmetic(edlib, mode, oracle):
    """every value within 1e-10 of the LIBM flavour (= the reference's arithmetic); paths and call tables as strict mode's"""
    E, S, C = 6000, 96, 5
    chrom_off, start, end = synth.exon_design(E, C, 2)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 2, n_segments=6, mean_depth=90.0)
    test[:40, :] = 0; ref[:40, :] = 0           # exons without reads
    test[100, :] = 0                            # obs = 0 over a deep reference
    ref[101, :] = 0                             # ref = 0
    plan = ed.Plan(chrom_off, start, end)
    r = run_modes(plan, S, test, ref, phi, p, mode)
    ll0, ll1 = r[0]["ll"], r[1]["ll"]
    assert np.all(close(ll1, ll0))
    assert np.all(ll1[:40] == 0.0) and not np.signbit(ll1[:40]).any()      # +0 exactly, as the reference's c - c
    worst = 0.0
    for s in range(0, S, 5):
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.LIBM)
        got = ll1[:, :, s]
        assert np.all(close(got, ell)), s
        nz = ell != 0
        worst = max(worst, float(np.max(np.abs(got[nz] - ell[nz]) / np.abs(ell[nz]))))
        epath, ecalls = oracle.callcnvs(ell, chrom_off, start, end)
        assert np.array_equal(r[1]["path"][:, s].astype(np.int8), epath), s
    assert worst < 1e-12, worst
    assert np.array_equal(r[0]["path"], r[1]["path"])
    assert np.array_equal(r[0]["calls"], r[1]["calls"])
    assert r[0]["nerr"] == r[1]["nerr"] == 0
    # the device-side tolerance check agrees
    v = r[1]["batch"].verify_emissions_tol(test, ref, phi, p, rel_tol=REL_TOL, abs_tol=ABS_TOL)
    assert v["compared"] == E * S * 3 and v["beyond"] == 0 and v["max_rel"] < 1e-12, v
    for m in (0, 1):
        r[m]["batch"].close()
    plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_cells_beyond_the_tables_carry_the_strict_bits(edlib, mode):
    """tiny tables: most cells are outside them and go through mode 0's arithmetic -- the list, and the full scan when it runs out"""
    E, S = 4000, 70
    chrom_off, start, end = synth.exon_design(E, 4, 3)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 3, n_segments=3, mean_depth=120.0)
    test[5, 3] = -4                      # a negative count: outside every table; NaN + error events as strict mode
    plan = ed.Plan(chrom_off, start, end)
    for reach, cap_obs, cap_ref in ((1.0, 64, 64), (1.0, 128, 1024), (8.0, 4096, 32768)):
        # (tails off: with tables this short against counts this deep the samples would be tail samples, served by the series wherever the tables end)
        r = run_modes(plan, S, test, ref, phi, p, mode, cap_obs=cap_obs, cap_ref=cap_ref, reach=reach, tails=0)
        b = r[1]["batch"]
        ll0, ll1 = r[0]["ll"], r[1]["ll"]
        out, notab = not_served(b, test, ref)
        assert not notab.any()
        st = b.table_stats()
        assert st["n_cold_cells"] == out.sum() == b.n_cold_cells(), (st, out.sum())     # summed over the launch groups, lists run out or not
        assert st["n_samples_without_tables"] == 0 and st["n_cells_without_tables"] == 0
        assert (st["cold_list_overflow"] > 0) if cap_ref == 64 else (st["cold_list_overflow"] == 0 or reach < 8.0), st
        sel = np.broadcast_to(out[:, None, :], ll0.shape)
        assert np.array_equal(bits(ll1[sel]), bits(ll0[sel]))                 # strict bits (NaN payloads included)
        assert np.all(close(ll1[~sel], ll0[~sel]))
        assert r[0]["nerr"] == r[1]["nerr"]
        assert np.array_equal(r[0]["path"], r[1]["path"]) and np.array_equal(r[0]["calls"], r[1]["calls"])
        for m in (0, 1):
            r[m]["batch"].close()
    plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_samples_the_tables_do_not_serve(edlib, mode):
    """phi >= 1 (negative shape parameters), expected outside (0, 1), a tiny expected (ill-conditioned sum), phi = 1e-9 (binomial: every cell
    under the few-reads rule), NaN: no tables for those samples -- every cell strict, bit for bit, error counts included, in one pass over
    the whole sample -- while their neighbours use theirs.  phi = 1e-4 and 3e-4 (nearly binomial) keep their tables and lose only the
    cells with few reads (the reference's own rounding noise is too close to the bar there)."""
    E, S = 1500, 24
    chrom_off, start, end = synth.exon_design(E, 2, 4)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 4, n_segments=2, mean_depth=60.0)
    phi = phi.copy(); p = p.copy()
    phi[1] = 1.5; phi[2] = 1.0; p[3] = 0.0; p[4] = 1.0; p[5] = 1e-7; phi[6] = np.nan; p[7] = -0.2; phi[8] = 0.0; phi[9] = 1e-9; phi[10] = 1e-4; phi[11] = 3e-4; phi[12] = 1e-3
    test[:30, 10] = [0, 1, 2] * 10; ref[:30, 10] = [1, 0, 3] * 10       # cells with 1 .. 5 reads in the nearly binomial sample
    plan = ed.Plan(chrom_off, start, end)
    r = run_modes(plan, S, test, ref, phi, p, mode)
    b = r[1]["batch"]
    ll0, ll1 = r[0]["ll"], r[1]["ll"]
    out, notab = not_served(b, test, ref)
    assert list(np.flatnonzero(notab)) == [1, 2, 3, 4, 5, 6, 7, 8, 9]
    assert b.table_dims(10)[2] >= 5 and b.table_dims(11)[2] >= 1 and b.table_dims(12)[2] == 0 and b.table_dims(0)[2] == 0
    assert out[:30, 10].all()
    for s in range(S):
        if notab[s]:
            assert np.array_equal(bits(ll1[:, :, s]), bits(ll0[:, :, s])), s
        else:
            assert np.all(close(ll1[:, :, s], ll0[:, :, s])), s
            assert np.array_equal(bits(ll1[out[:, s], :, s]), bits(ll0[out[:, s], :, s])), s
    st = b.table_stats()
    assert st["n_samples_without_tables"] == 9 and st["n_cells_without_tables"] == 9 * E and st["cold_list_overflow"] == 0, st
    assert st["n_cold_cells"] == out[:, ~notab].sum(), st
    assert r[0]["nerr"] == r[1]["nerr"] and r[0]["nerr"] > 0
    assert np.array_equal(r[0]["path"], r[1]["path"]) and np.array_equal(r[0]["calls"], r[1]["calls"])
    for m in (0, 1):
        r[m]["batch"].close()
    plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_every_sample_without_tables(edlib, mode):
    """a batch none of whose samples the tables serve (phi >= 1): the strict pass walks all of it -- mode 0's bits, no list entries, no re-scan"""
    E, S = 2600, 80
    chrom_off, start, end = synth.exon_design(E, 3, 14)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 14, n_segments=2, mean_depth=40.0)
    phi = np.full(S, 1.25)
    plan = ed.Plan(chrom_off, start, end)
    for overlap in (0, 1):
        r = {}
        for key, m in ((0, 0), (1, mode)):
            b = ed.Batch(plan, S); b.set_viterbi_overlap(overlap)
            if m:
                b.set_emit_mode(m)
            b.run(test, ref, phi, p)
            r[key] = dict(ll=b.loglik(), path=b.path(), calls=b.calls(), nerr=b.n_gsl_errors(), batch=b)
        assert np.array_equal(bits(r[1]["ll"]), bits(r[0]["ll"]))
        st = r[1]["batch"].table_stats()
        assert st == {"n_cold_cells": 0, "n_samples_without_tables": S, "cold_list_overflow": 0, "n_cells_without_tables": S * E}, st
        assert r[0]["nerr"] == r[1]["nerr"]
        assert np.array_equal(r[0]["path"], r[1]["path"]) and np.array_equal(r[0]["calls"], r[1]["calls"])
        for m in (0, 1):
            r[m]["batch"].close()
    plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_reference_rounds_its_second_argument(edlib, oracle, mode):
    """expected close to 1 (the test sample far deeper than its reference) makes a2 << 1; the reference forms its second argument as
    (a2 + total) - observed (src/CNV_estimate.cpp:49), rounded at the size of the total -- for ref = 0 its own value is then 2e-10 away
    from the exact one (found by tools/fuzz_tables.py: phi 0.1988, expected 0.99907, (obs, ref) = (24, 0), duplication state), while the
    tables are exact to 2e-12.  Parity is measured against the reference: those cells take its arithmetic (the N0 of table_dims)."""
    rng = np.random.default_rng(8)
    E, S = 1200, 12
    chrom_off, start, end = synth.exon_design(E, 2, 8)
    p = np.array([0.9990708643096986] * 4 + [0.998, 0.9995, 0.99, 0.97] * 2)
    phi = np.array([0.19884239345741983] * 4 + [0.05, 0.3, 0.01, 0.1, 0.02, 0.15, 0.2, 0.005])
    tot = rng.poisson(np.exp(rng.uniform(0, np.log(3000.0), (E, S)))).astype(np.int64)
    ref = rng.binomial(tot, (1 - p)[None, :]).astype(np.int32)
    test = (tot - ref).astype(np.int32)
    test[0, 0] = 24; ref[0, 0] = 0
    plan = ed.Plan(chrom_off, start, end)
    r = run_modes(plan, S, test, ref, phi, p, mode)
    b = r[1]["batch"]
    out, notab = not_served(b, test, ref)
    assert out[0, 0]            # (that sample: a2 = 1.5e-3 -- no tables at all; the rule below is for the a2 of a few hundredths)
    n0 = np.array([b.table_dims(s)[3] if not notab[s] else 2**31 - 1 for s in range(S)])
    by_rule = sum(int(np.sum((ref[:, s] == 0) & (test[:, s] >= n0[s]) & (test[:, s] < b.table_dims(s)[0]))) for s in range(S) if not notab[s])
    assert (n0 < 2**31 - 1).any() and by_rule > 0, (n0, by_rule)
    for s in range(S):
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.LIBM)
        assert np.all(close_rel(r[1]["ll"][:, :, s], ell)), (s, phi[s], p[s])
    assert np.all(close_rel(r[1]["ll"], r[0]["ll"]))
    sel = np.broadcast_to(out[:, None, :], r[0]["ll"].shape)
    assert np.array_equal(bits(r[1]["ll"][sel]), bits(r[0]["ll"][sel]))
    assert np.array_equal(r[0]["path"], r[1]["path"]) and np.array_equal(r[0]["calls"], r[1]["calls"])
    for m in (0, 1):
        r[m]["batch"].close()
    plan.close()


def aggregate_reference_batch(E, S, seed, depth=100.0, kmin=20, kmax=32):
    """the reference's workflow in small: a cohort's counts; every sample's reference is the sum of 20 - 32 OTHER samples of the cohort
    (vignette/vignette.Rnw:390-402), i.e. deep references, expected ~ 1 / (k + 1) = 0.03 - 0.05"""
    chrom_off, start, end = synth.exon_design(E, 4, seed)
    cohort, _, _, _, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=3, mean_depth=depth)
    rng = np.random.default_rng(seed)
    ref = np.zeros_like(cohort)
    for s in range(S):
        k = int(rng.integers(kmin, kmax + 1))
        others = rng.choice(np.delete(np.arange(S), s), size=min(k, S - 1), replace=False)
        ref[:, s] = cohort[:, others].sum(axis=1)
    return chrom_off, start, end, cohort, ref


@pytest.mark.parametrize("mode,layout", [(2, 1), (2, 0), (1, 0)])
def test_deep_aggregate_references(edlib, oracle, mode, layout):
    """references that are 20 - 32-sample aggregates (the reference's real usage).  Dispersion fitted on the device;
    >= 95 % of the samples on tables, the strict lists a small fraction of the cells, every log-likelihood within 1e-10 RELATIVE of the
    reference's arithmetic (no absolute floor), 0 discordant Viterbi states / call rows against it."""
    E, S = 12_000, 72
    chrom_off, start, end, test, ref = aggregate_reference_batch(E, S, 31)
    test[:25, :] = 0; ref[:25, :] = 0            # exons without reads
    test[40:60, :] = 0                           # obs = 0 over a deep reference: the smallest values of a column
    plan = ed.Plan(chrom_off, start, end)
    b = ed.Batch(plan, S)
    b.set_emit_mode(mode); b.set_counts_layout(layout)
    t_in, r_in = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
    dt, dr = ed.DeviceArray(t_in), ed.DeviceArray(r_in)
    dphi, dexp = ed.DeviceArray(np.zeros(S)), ed.DeviceArray(np.zeros(S))
    b.fit(dt, dr, dphi, dexp)
    b.run(dt, dr, dphi, dexp)
    phi, p = np.asarray(dphi.to_host()), np.asarray(dexp.to_host())
    assert np.all((p > 0.015) & (p < 0.1)) and np.median(p) < 0.05, (p.min(), p.max())
    st = b.table_stats()
    out, notab = not_served(b, test, ref)
    assert notab.sum() <= S // 20, (notab.sum(), st)
    assert st["n_samples_without_tables"] == notab.sum() and st["cold_list_overflow"] == 0, st
    assert st["n_cold_cells"] == out[:, ~notab].sum() <= E * S // 200, st
    ll, path, calls = b.loglik(), b.path(), b.calls()
    n_beyond = n_floor = n_states = n_rows = 0
    worst = 0.0
    for s in range(S):
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.LIBM)
        got = ll[:, :, s]
        n_beyond += int(np.sum(~close_rel(got, ell)))
        n_floor += int(np.sum(close(got, ell) & ~close_rel(got, ell)))
        nz = np.isfinite(ell) & (ell != 0)
        worst = max(worst, float(np.max(np.abs(got[nz] - ell[nz]) / np.abs(ell[nz]))))
        epath, ecalls = oracle.callcnvs(ell, chrom_off, start, end)
        n_states += int(np.sum(path[:, s].astype(np.int8) != epath))
        mine = calls[calls["sample"] == s]
        gotc = {tuple(int(v) for v in row) for row in zip(mine["start_exon"] + 1, mine["end_exon"] + 1, mine["type"], mine["nexons"])}
        n_rows += len(gotc ^ {tuple(int(v) for v in row[:4]) for row in ecalls})
    assert n_beyond == 0 and n_floor == 0, (n_beyond, n_floor, worst)
    assert n_states == 0 and n_rows == 0, (n_states, n_rows)
    assert worst < 2e-11, worst
    b.close(); plan.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("S", [1, 15, 16, 17, 130])
def test_ragged_sample_blocks_and_schedules(edlib, S, mode):
    """sample counts around the tile width; one emission launch per batch and the overlap groups; a second run on the same batch"""
    E = 2500
    chrom_off, start, end = synth.exon_design(E, 6, 9)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 9, n_segments=3, mean_depth=80.0)
    plan = ed.Plan(chrom_off, start, end)
    ref_b = ed.Batch(plan, S)
    ref_b.run(test, ref, phi, p)
    ll0, path0, calls0 = ref_b.loglik(), ref_b.path(), ref_b.calls()
    for overlap in (1, 0):
        b = ed.Batch(plan, S)
        b.set_viterbi_overlap(overlap)
        b.set_emit_mode(mode)
        for _ in range(2):
            b.run(test, ref, phi, p)
            assert np.all(close(b.loglik(), ll0))
            assert np.array_equal(b.path(), path0) and np.array_equal(b.calls(), calls0)
        b.close()
    ref_b.close(); plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_deep_and_shallow_counts(edlib, mode, oracle):
    """mean depths 3 and 1500 reads per exon: short tables, and tables at their caps with a tail through the strict arithmetic"""
    E, S = 3000, 32
    chrom_off, start, end = synth.exon_design(E, 3, 6)
    plan = ed.Plan(chrom_off, start, end)
    for depth in (3.0, 1500.0):
        test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 6, n_segments=3, mean_depth=depth)
        r = run_modes(plan, S, test, ref, phi, p, mode)
        assert np.all(close(r[1]["ll"], r[0]["ll"]))
        for s in (0, S - 1):
            ell, _ = oracle.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.LIBM)
            assert np.all(close(r[1]["ll"][:, :, s], ell))
        assert np.array_equal(r[0]["path"], r[1]["path"]) and np.array_equal(r[0]["calls"], r[1]["calls"])
        for m in (0, 1):
            r[m]["batch"].close()
    plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_wide_parameter_grid(edlib, mode, oracle):
    """phi from 1e-6 to 0.9, expected from 1e-3 to 0.999: whatever gets tables is within 1e-10 of the reference's arithmetic,
    whatever does not carries the strict bits"""
    rng = np.random.default_rng(21)
    E, S = 800, 64
    chrom_off, start, end = synth.exon_design(E, 2, 8)
    phi = np.exp(rng.uniform(np.log(1e-6), np.log(0.9), S))
    p = np.concatenate([np.exp(rng.uniform(np.log(1e-3), np.log(0.5), S // 2)), 1 - np.exp(rng.uniform(np.log(1e-3), np.log(0.5), S - S // 2))])
    tot = rng.poisson(np.exp(rng.uniform(0, np.log(3000.0), (E, S)))).astype(np.int64)
    test = rng.binomial(tot, p[None, :]).astype(np.int32)
    ref = (tot - test).astype(np.int32)
    plan = ed.Plan(chrom_off, start, end)
    r = run_modes(plan, S, test, ref, phi, p, mode)
    b = r[1]["batch"]
    n_tab = 0
    for s in range(S):
        ly, lr = b.emit_tables(s)[:2]
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.LIBM)
        assert np.all(close(r[1]["ll"][:, :, s], ell)), (s, phi[s], p[s], ly, lr)
        if ly == 0:
            assert np.array_equal(bits(r[1]["ll"][:, :, s]), bits(r[0]["ll"][:, :, s]))
        n_tab += ly > 0
    assert n_tab >= S // 4, n_tab
    for m in (0, 1):
        r[m]["batch"].close()
    plan.close()


def test_sample_major_counts(edlib):
    """ed_batch_set_counts_layout(1): the counts as [n_samples][n_exons] -- R's column-major matrix as it lies in memory -- through
    the histogram fit and emit mode 2 without a transposition: the same fit (to its tolerance), and given the same parameters the
    same likelihood bits, paths, calls and decoration as the [n_exons][n_samples] entry"""
    E, S = 5000, 100
    chrom_off, start, end = synth.exon_design(E, 5, 12)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 12, n_segments=4, mean_depth=110.0)
    test[7, 2] = -1
    tt, rt = np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)
    plan = ed.Plan(chrom_off, start, end)
    a = ed.Batch(plan, S); a.set_emit_mode(2)
    b = ed.Batch(plan, S); b.set_emit_mode(2); b.set_counts_layout(1)
    fa = [ed.DeviceArray(np.zeros(S)) for _ in range(2)]
    fb = [ed.DeviceArray(np.zeros(S)) for _ in range(2)]
    a.fit(test, ref, fa[0], fa[1]); b.fit(tt, rt, fb[0], fb[1])
    a.run(test, ref, phi, p); b.run(tt, rt, phi, p)
    assert np.array_equal(bits(a.loglik()), bits(b.loglik()))
    assert np.array_equal(a.path(), b.path()) and np.array_equal(a.calls(), b.calls())
    assert a.n_gsl_errors() == b.n_gsl_errors()
    ia, ib = a.call_info(), b.call_info()
    assert np.array_equal(ia, ib)
    for x, y in zip(fa, fb):
        x, y = x.to_host(), y.to_host()
        assert np.all(np.abs(x - y) <= 1e-9 * np.abs(x)), np.max(np.abs(x - y) / np.abs(x))
    assert a.fit_unconverged()[0] == b.fit_unconverged()[0] == 0
    # the device-side verification reads the counts in the batch's layout
    v = b.verify_emissions_tol(tt, rt, phi, p, rel_tol=REL_TOL, abs_tol=ABS_TOL)
    assert v["compared"] == E * S * 3 and v["beyond"] == 0, v
    # fit mode 1 (Nelder-Mead on the same histograms) and a strided fit (subset.for.speed)
    for m in (a, b):
        from exomedepth_amd._lib import check, lib
        check(lib().ed_batch_set_fit_mode(m.handle, 1))
    a.fit(test, ref, fa[0], fa[1]); b.fit(tt, rt, fb[0], fb[1])
    assert np.all(np.abs(fa[0].to_host() - fb[0].to_host()) <= 5e-3 * fa[0].to_host())
    for m in (a, b):
        check(lib().ed_batch_set_fit_mode(m.handle, 0))
    a.fit(test, ref, fa[0], fa[1], by=7); b.fit(tt, rt, fb[0], fb[1], by=7)
    assert np.all(np.abs(fa[0].to_host() - fb[0].to_host()) <= 1e-8 * fa[0].to_host())
    # modes that do not take the layout say so
    c = ed.Batch(plan, S); c.set_counts_layout(1)
    with pytest.raises(Exception, match="emit mode 2"):
        c.run(tt, rt, phi, p)
    for m in (a, b, c):
        m.close()
    plan.close()


@pytest.mark.parametrize("E,depth", [(5000, 110.0), (4096 * 3 + 17, 25.0), (70000, 400.0)])
def test_sixteen_bit_counts(edlib, E, depth):
    """ed_batch_set_counts_bits(16): uint16 [n_samples][n_exons] device counts through the sample-major table mode -- moments, histograms (16-byte loads of
    eight counts, spans of 2 048 exons, the scalar remainder), table statistics, emissions, the strict pass for cells outside the tables, the decoration: the
    same likelihood bits, paths, calls and decoration as the int32 form given the same parameters; the same fit to its tolerance (a sample's overflow cells
    are met in another order); through the cohort pipeline too.  Counts that do not fit, other modes and host-fed slabs say so."""
    S = 96
    chrom_off, start, end = synth.exon_design(E, 5, 21)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 21, n_segments=4, mean_depth=depth)
    test[5, 3] = 0; ref[5, 3] = 0
    test[11, :] = test[11, :] * 40                                     # cells beyond the tables (the strict list)
    test, ref = np.minimum(test, 65535), np.minimum(ref, 65535)        # (what the format holds)
    tt, rt = np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)
    t16, r16 = tt.astype(np.uint16), rt.astype(np.uint16)
    assert np.array_equal(t16.astype(np.int32), tt) and np.array_equal(r16.astype(np.int32), rt)
    plan = ed.Plan(chrom_off, start, end)
    a = ed.Batch(plan, S); a.set_emit_mode(2); a.set_counts_layout(1)
    b = ed.Batch(plan, S); b.set_emit_mode(2); b.set_counts_layout(1); b.set_counts_bits(16)
    fa = [ed.DeviceArray(np.zeros(S)) for _ in range(2)]
    fb = [ed.DeviceArray(np.zeros(S)) for _ in range(2)]
    a.fit(tt, rt, fa[0], fa[1]); b.fit(t16, r16, fb[0], fb[1])
    for x, y in zip(fa, fb):
        x, y = x.to_host(), y.to_host()
        assert np.all(np.abs(x - y) <= 1e-9 * np.abs(x)), np.max(np.abs(x - y) / np.abs(x))
    assert a.fit_unconverged()[0] == b.fit_unconverged()[0] == 0
    a.run(tt, rt, phi, p); b.run(t16, r16, phi, p)
    assert np.array_equal(bits(a.loglik()), bits(b.loglik()))
    assert np.array_equal(a.path(), b.path()) and np.array_equal(a.calls(), b.calls()) and len(a.calls()) > 0
    assert np.array_equal(a.call_info(), b.call_info())
    assert a.n_gsl_errors() == b.n_gsl_errors() and a.table_stats() == b.table_stats() and a.table_stats()["n_cold_cells"] > 0
    v = b.verify_emissions_tol(t16, r16, phi, p, rel_tol=REL_TOL, abs_tol=ABS_TOL)          # the device-side check reads the 16-bit counts too
    assert v["compared"] == E * S * 3 and v["beyond"] == 0, v
    # fitted on the device and run with what was fitted: the same calls (the parameters agree to 1e-9)
    a.run(tt, rt, fa[0], fa[1]); b.run(t16, r16, fb[0], fb[1])
    assert np.array_equal(a.path(), b.path()) and np.array_equal(a.calls(), b.calls())
    # strided fit and fit mode 1
    a.fit(tt, rt, fa[0], fa[1], by=7); b.fit(t16, r16, fb[0], fb[1], by=7)
    assert np.all(np.abs(fa[0].to_host() - fb[0].to_host()) <= 1e-8 * fa[0].to_host())
    from exomedepth_amd._lib import check, lib
    for m in (a, b):
        check(lib().ed_batch_set_fit_mode(m.handle, 1))
    a.fit(tt, rt, fa[0], fa[1]); b.fit(t16, r16, fb[0], fb[1])
    assert np.all(np.abs(fa[0].to_host() - fb[0].to_host()) <= 5e-3 * fa[0].to_host())
    # through the cohort pipeline (device slabs)
    want_calls, want_path = None, None
    for bits_ in (32, 16):
        co = ed.Cohort(plan, S, 2, emit_mode=2, counts_layout=1, counts_bits=bits_)
        tk = co.submit(ed.DeviceArray(tt if bits_ == 32 else t16), ed.DeviceArray(rt if bits_ == 32 else r16), n_samples=S)
        got = co.results(tk, S, path=True)
        if want_calls is None:
            want_calls, want_path = got["calls"], got["path"]
        else:
            assert np.array_equal(got["calls"], want_calls) and np.array_equal(got["path"], want_path)
            with pytest.raises(Exception, match="counts_bits"):
                co.submit_host(t16, r16, 1)
        co.close()
    # what does not take the format says so
    with pytest.raises(ValueError, match="65535"):
        b.run(tt + 70000, rt, phi, p)
    c = ed.Batch(plan, S); c.set_emit_mode(2); c.set_counts_bits(16)
    with pytest.raises(Exception, match="counts_layout 1"):
        c.run(t16, r16, phi, p)
    with pytest.raises(Exception, match="sample-major"):
        c.fit(t16, r16, fb[0], fb[1])
    for m in (a, b, c):
        m.close()
    plan.close()


def test_cohort_pipeline_in_table_modes(edlib):
    """slabs through the cohort pipeline (two in flight) in emit modes 1 and 2, device counts in both layouts and host-fed slabs in R's
    layout: the calls of the strict pipeline, the fitted parameters to the fit's tolerance"""
    E, S, slab = 3000, 96, 32
    chrom_off, start, end = synth.exon_design(E, 4, 13)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 13, n_segments=4, mean_depth=90.0)
    plan = ed.Plan(chrom_off, start, end)

    def run(**opts):
        co = ed.Cohort(plan, slab, 2, **opts)
        lay = opts.get("counts_layout", 0)
        out = []
        tickets = []
        hold = []
        for s0 in range(0, S, slab):
            t, r = test[:, s0:s0 + slab], ref[:, s0:s0 + slab]
            t = np.ascontiguousarray(t.T if lay else t); r = np.ascontiguousarray(r.T if lay else r)
            hold.append((ed.DeviceArray(t), ed.DeviceArray(r)))
            tickets.append(co.submit(hold[-1][0], hold[-1][1], n_samples=slab))
            if len(tickets) >= 2:
                out.append(co.results(tickets[-2], slab, path=True))
        out.append(co.results(tickets[-1], slab, path=True))
        co.close()
        return out

    base = run()
    for opts in ({"emit_mode": 1}, {"emit_mode": 2}, {"emit_mode": 2, "counts_layout": 1}):
        got = run(**opts)
        for g, w in zip(got, base):
            assert np.array_equal(g["path"], w["path"]), opts
            assert np.array_equal(g["calls"], w["calls"]), opts
            assert np.all(np.abs(g["phi"] - w["phi"]) <= 1e-9 * w["phi"]), opts
    # host-fed, R's layout, both wire formats, straight into the sample-major pipeline
    for wire_dtype in (np.int32, np.uint16):
        co = ed.Cohort(plan, slab, 2, emit_mode=2, counts_layout=1)
        res = co.run_host(np.ascontiguousarray(test.T.astype(wire_dtype)), np.ascontiguousarray(ref.T.astype(wire_dtype)), layout=1, want_path=True)
        co0 = ed.Cohort(plan, slab, 2)
        res0 = co0.run_host(np.ascontiguousarray(test.T.astype(wire_dtype)), np.ascontiguousarray(ref.T.astype(wire_dtype)), layout=1, want_path=True)
        assert np.array_equal(res["path"], res0["path"]) and np.array_equal(res["calls"], res0["calls"])
        co.close(); co0.close()
    plan.close()


def test_bundled_data_in_table_modes(edlib, oracle):
    """BASELINE.json configs[0]: the four samples of the reference's bundled data (tests/golden/exomecount_chr1.npz) at the parameters the
    reference's arithmetic fits (stored in the fixture), in both table modes: 0 discordant Viterbi states / call rows against the stored
    result of the libm flavour, log-likelihoods within the tolerance of it"""
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = np.load(os.path.join(G, "exomecount_chr1.npz"))
    exp = np.load(os.path.join(G, "config1_expected.npz"))
    start, end, counts = d["start"], d["end"], d["counts"].astype(np.int32)
    n = start.size
    chrom_off = np.array([0, n], np.int32)
    test = np.ascontiguousarray(counts[:, :4])
    ref = np.ascontiguousarray(counts.sum(axis=1, dtype=np.int32)[:, None] - test)
    phi = np.array([float(exp["phi%d" % i]) for i in range(4)])
    p = np.array([float(exp["p%d" % i]) for i in range(4)])
    plan = ed.Plan(chrom_off, start, end)
    for mode, layout in ((1, 0), (2, 0), (2, 1)):
        b = ed.Batch(plan, 4)
        b.set_emit_mode(mode); b.set_counts_layout(layout)
        b.run(*((np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)), phi, p)
        ll, path, calls = b.loglik(), b.path(), b.calls()
        for i in range(4):
            assert b.emit_tables(i)[0] > 0
            ell, _ = oracle.get_loglike_matrix(phi[i], p[i], test[:, i] + ref[:, i], test[:, i], 1.0, oracle.LIBM)
            assert close(ll[:, :, i], ell).all()
            assert np.array_equal(path[:, i].astype(np.int8), exp["path%d" % i]), (mode, layout, i)
            mine = calls[calls["sample"] == i]
            got = np.stack([mine["start_exon"] + 1, mine["end_exon"] + 1, mine["type"], mine["nexons"]], axis=1).astype(np.int64)
            assert np.array_equal(got, exp["calls%d" % i].astype(np.int64)), (mode, layout, i)
        b.close()
    plan.close()


def _whole_columns_against_libm(oracle, plan, chrom_off, start, end, test, ref, phi, p, path, calls, ll, cols):
    """per column: (values beyond tolerance, discordant states, discordant call rows) against the checker's libm flavour"""
    from concurrent.futures import ThreadPoolExecutor

    def one(s):
        t, r = test[:, s], ref[:, s]
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], t + r, t, 1.0, oracle.LIBM)
        epath, ecalls = oracle.callcnvs(ell, chrom_off, start, end)
        mine = calls[calls["sample"] == s]
        got = {tuple(int(v) for v in row) for row in zip(mine["start_exon"] + 1, mine["end_exon"] + 1, mine["type"], mine["nexons"])}
        want = {tuple(int(v) for v in row[:4]) for row in ecalls}
        return int(np.sum(~close(ll[:, :, s], ell))), int(np.sum(path[:, s].astype(np.int8) != epath)), len(got ^ want)
    import os
    with ThreadPoolExecutor(max(1, min(32, (os.cpu_count() or 2) - 1))) as ex:     # the checker is a C call: the GIL is released
        return list(ex.map(one, cols))


@pytest.mark.parametrize("S,mode,layout", [(64, 2, 1), (64, 1, 0), (1024, 2, 1)])
def test_whole_columns_of_the_headline_geometries(edlib, oracle, S, mode, layout):
    """BASELINE.json configs[1] (200 000 x 64, every column) and configs[2] (200 000 x 1024, 64 columns spread over the batch), dispersion
    fitted on the device, emissions from the tables: log-likelihoods within 1e-10 of the reference's arithmetic and 0 discordant Viterbi
    states / call rows against it on every one of those 200 000-exon columns"""
    E, C = 200_000, 24
    chrom_off, start, end = synth.exon_design(E, C, seed=20250620)
    test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, 20250627, n_segments=6, mean_depth=100.0)
    plan = ed.Plan(chrom_off, start, end)
    b = ed.Batch(plan, S)
    b.set_emit_mode(mode); b.set_counts_layout(layout)
    t_in, r_in = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
    dt, dr = ed.DeviceArray(t_in), ed.DeviceArray(r_in)
    dphi, dexp = ed.DeviceArray(np.zeros(S)), ed.DeviceArray(np.zeros(S))
    b.fit(dt, dr, dphi, dexp)
    b.run(dt, dr, dphi, dexp)
    phi, p = dphi.to_host(), dexp.to_host()
    assert b.fit_unconverged()[0] == 0
    n_tab = sum(1 for s in range(S) if b.emit_tables(s)[0] > 0)
    assert n_tab == S
    cols = list(range(S)) if S == 64 else [int(v) for v in np.linspace(0, S - 1, 64).round()]
    path, calls = b.path(), b.calls()
    ll = b.loglik()
    res = _whole_columns_against_libm(oracle, plan, chrom_off, start, end, test, ref, np.asarray(phi), np.asarray(p), path, calls, ll, cols)
    assert sum(r[0] for r in res) == 0, "log-likelihoods beyond the tolerance"
    assert sum(r[1] for r in res) == 0, "discordant Viterbi states"
    assert sum(r[2] for r in res) == 0, "discordant call rows"
    b.close(); plan.close()


# ---- tail samples (round 6): counts that outgrow the LDS windows of the sample-major form are served by Stirling's series (csrc/ed_dtab.h) ----------
def _tail_check(oracle, b, test, ref, phi, p, plan, S, strict, expect_tail, beyond_the_limit=False):
    ll = b.loglik()
    wins = [b.table_windows(s) for s in range(S)]
    assert [w[3] for w in wins] == list(expect_tail), [w[3] for w in wins]
    st = b.table_stats()
    out, _ = not_served(b, test, ref)
    assert st["n_samples_without_tables"] == 0 and st["n_cold_cells"] == int(out.sum())
    if not beyond_the_limit:
        assert st["cold_list_overflow"] == 0
        assert out.mean() < 0.02, out.mean()                      # the series serves up to the conditioning limit: next to nothing is left to the strict lists
    sel = np.broadcast_to(out[:, None, :], ll.shape)
    assert np.array_equal(bits(ll[sel]), bits(strict["ll"][sel]))     # ... and what is carries mode 0's bits
    worst = 0.0
    for s in range(S):
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.LIBM)
        assert np.all(close_rel(ll[:, :, s], ell)), s
        nz = ell != 0
        worst = max(worst, float(np.max(np.abs(ll[:, :, s][nz] - ell[nz]) / np.abs(ell[nz]))))
    assert np.array_equal(b.path(), strict["path"]) and np.array_equal(b.calls(), strict["calls"])
    return wins, worst


@pytest.mark.parametrize("depth", [400.0, 1600.0, 6000.0])
@pytest.mark.parametrize("layout", [0, 1])
def test_tail_samples_at_depth(edlib, oracle, depth, layout):
    """every sample a tail sample: windows in LDS, the series beyond, values within 1e-10 RELATIVE of the reference's arithmetic, the strict
    mode's paths and calls; with the tails switched off (round 5's full-length tables) the same paths and calls"""
    E, S = 6000, 40
    chrom_off, start, end = synth.exon_design(E, 4, 31)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 31, n_segments=5, mean_depth=depth)
    plan = ed.Plan(chrom_off, start, end)
    b0 = ed.Batch(plan, S); b0.run(test, ref, phi, p)
    strict = dict(ll=b0.loglik(), path=b0.path(), calls=b0.calls())
    t_in, r_in = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
    b = ed.Batch(plan, S); b.set_emit_mode(2); b.set_counts_layout(layout)
    b.run(t_in, r_in, phi, p)
    # (6 000 reads per exon: totals of 50 000 and more, beyond where the conditioning rule lets the three-term sum stand in for the reference's value --
    #  those cells take the strict arithmetic, lists run out or not)
    wins, worst = _tail_check(oracle, b, test, ref, phi, p, plan, S, strict, [1] * S, beyond_the_limit=depth > 2000)
    assert all(w[0] + w[1] + w[2] <= 6144 and min(w[:3]) >= 64 for w in wins)
    ly, lr, t1, t2, t3 = b.emit_tables(0)                          # a tail sample's tables ARE its windows
    assert (ly, lr) == wins[0][:2]
    assert worst < 5e-12, worst
    b2 = ed.Batch(plan, S); b2.set_emit_mode(2, tails=0); b2.set_counts_layout(layout)
    b2.run(t_in, r_in, phi, p)
    assert all(b2.table_windows(s)[3] == 0 for s in range(S))
    assert np.array_equal(b2.path(), strict["path"]) and np.array_equal(b2.calls(), strict["calls"])
    for x in (b0, b, b2):
        x.close()
    plan.close()


def test_tail_and_table_samples_in_one_slab(edlib, oracle):
    """shallow and deep samples side by side (70 and 1 500 reads per exon), 130 of them (ragged against every tile width), fitted on the device"""
    E, S = 5000, 130
    chrom_off, start, end = synth.exon_design(E, 3, 32)
    deep = (np.arange(S) % 3 == 1)
    ta, ra, pa, pha, _ = synth.counts_numpy(chrom_off, S, 32, n_segments=4, mean_depth=70.0)
    tb_, rb_, pb, phb, _ = synth.counts_numpy(chrom_off, S, 33, n_segments=4, mean_depth=1500.0)
    test = np.where(deep[None, :], tb_, ta).astype(np.int32); ref = np.where(deep[None, :], rb_, ra).astype(np.int32)
    p = np.where(deep, pb, pa); phi = np.where(deep, phb, pha)
    plan = ed.Plan(chrom_off, start, end)
    b0 = ed.Batch(plan, S); b0.run(test, ref, phi, p)
    strict = dict(ll=b0.loglik(), path=b0.path(), calls=b0.calls())
    b = ed.Batch(plan, S); b.set_emit_mode(2); b.set_counts_layout(1)
    b.run(np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T), phi, p)
    _tail_check(oracle, b, test, ref, phi, p, plan, S, strict, deep.astype(int))
    b.run(np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T), phi, p)       # the object is reused: the tail list is rebuilt
    _tail_check(oracle, b, test, ref, phi, p, plan, S, strict, deep.astype(int))
    b0.close(); b.close(); plan.close()


def test_tail_values_are_the_checkers(edlib, oracle):
    """a tail sample's emissions, recomputed on the host from the shared definition: table entries inside the windows (edo_dtab), ed_dtab_tail beyond
    them, ed_dtab_combine of the three -- the same bits (a table entry on a rounding boundary may sit an ulp off: the parallel scan's association)"""
    E, S = 3000, 6
    chrom_off, start, end = synth.exon_design(E, 2, 34)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 34, n_segments=2, mean_depth=900.0)
    plan = ed.Plan(chrom_off, start, end)
    b = ed.Batch(plan, S); b.set_emit_mode(2)
    b.run(test, ref, phi, p)
    ll = b.loglik()
    n_eq = n_all = 0
    for s in (0, S - 1):
        n1, n2, n3, tail = b.table_windows(s)
        assert tail == 1
        out, _ = not_served(b, test, ref)
        e = p[s]
        sd = np.sqrt((phi[s] * e) * (1.0 - e))
        for st, odds in enumerate((0.5, 1.0, 1.5)):
            ep = e if st == 1 else (e * odds) / ((e * odds + 1) - e)
            a1 = ((ep * ep) * (1 - ep)) / (sd * sd) - ep
            a2 = ((1 - ep) / ep) * a1
            parts = []
            for x0, idx, nwin in ((a1, test[:, s], n1), (a2, ref[:, s], n2), (a1 + a2, test[:, s] + ref[:, s], n3)):
                tab = oracle.dtab(x0, nwin)
                inside = idx < nwin
                d = np.empty(E)
                d[inside] = tab[idx[inside]]
                d[~inside] = oracle.dtab_tail(x0, idx[~inside].astype(np.float64))
                parts.append(d)
            want = oracle.dtab_combine(*parts)
            ok = ~out[:, s]
            got = ll[ok, st, s]
            n_eq += int(np.sum(bits(got) == bits(want[ok]))); n_all += int(ok.sum())
            assert np.all(np.abs(got - want[ok]) <= 4 * np.spacing(np.abs(want[ok])))
    assert n_eq >= 0.995 * n_all, (n_eq, n_all)
    b.close(); plan.close()
