"""Parity at BASELINE.json's full size (200 000 exons x 1024 samples on one GPU): oracle spot checks on
whole sample columns plus size-independent properties (determinism, independence of samples, call table
<-> path consistency)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E, S, C = 200_000, 1024, 24


@pytest.fixture(scope="module")
def full(edlib):
    torch = pytest.importorskip("torch")
    from exomedepth_amd import synth
    dev = torch.device("cuda", 0)
    chrom_off, start, end = synth.exon_design(E, C, seed=20250620)
    torch.manual_seed(7)
    test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=20250627)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    phi_f = torch.empty(S, dtype=torch.float64, device=dev)
    p_f = torch.empty(S, dtype=torch.float64, device=dev)
    batch.fit(test, ref, phi_f, p_f)
    batch.run(test, ref, phi_f, p_f)
    out = dict(edlib=edlib, torch=torch, chrom_off=chrom_off, start=start, end=end, test=test, ref=ref, plan=plan,
               batch=batch, phi=phi_f, p=p_f, path=batch.path(), calls=batch.calls())
    yield out
    batch.close(); plan.close()


def test_oracle_spot_check_whole_columns(full, oracle):
    """Three whole sample columns (200 000 exons each) against the checker: log-likelihoods through a
    one-column batch bit for bit, Viterbi path and call table identical."""
    phi = full["phi"].cpu().numpy(); p = full["p"].cpu().numpy()
    for s in (0, 517, 1023):
        t = full["test"][:, s].cpu().numpy(); r = full["ref"][:, s].cpu().numpy()
        ell, nerr = oracle.get_loglike_matrix(phi[s], p[s], t + r, t, 1.0, oracle.PORTABLE)
        assert nerr == 0
        epath, ecalls = oracle.callcnvs(ell, full["chrom_off"], full["start"], full["end"])
        assert np.array_equal(full["path"][:, s].astype(np.int8), epath)
        mine = full["calls"][full["calls"]["sample"] == s]
        assert np.array_equal(mine["start_exon"] + 1, ecalls[:, 0].astype(np.int64))
        assert np.array_equal(mine["end_exon"] + 1, ecalls[:, 1].astype(np.int64))
        assert np.array_equal(mine["type"], ecalls[:, 2].astype(np.int64))
        assert np.array_equal(mine["nexons"], ecalls[:, 3].astype(np.int64))
        # likelihood of that column through the device, as its own batch (independence of samples)
        b1 = full["edlib"].Batch(full["plan"], 1)
        b1.run(t.reshape(-1, 1), r.reshape(-1, 1), phi[s:s + 1], p[s:s + 1])
        assert np.array_equal(b1.loglik()[:, :, 0].view(np.int64), np.ascontiguousarray(ell).view(np.int64))
        assert np.array_equal(b1.path()[:, 0], full["path"][:, s])
        b1.close()
        # level C concordance against the reference's own arithmetic (libm flavour): discordant cells
        lll, _ = oracle.get_loglike_matrix(phi[s], p[s], t + r, t, 1.0, oracle.LIBM)
        lpath, _ = oracle.callcnvs(lll, full["chrom_off"], full["start"], full["end"])
        assert int(np.sum(lpath != epath)) == 0


def test_deterministic_rerun(full):
    h1 = hashlib.sha256(full["path"].tobytes()).hexdigest()
    full["batch"].run(full["test"], full["ref"], full["phi"], full["p"])
    assert hashlib.sha256(full["batch"].path().tobytes()).hexdigest() == h1
    assert np.array_equal(full["batch"].calls(), full["calls"])
    # the fit is reproducible too (fixed-order reductions)
    torch = full["torch"]
    phi2 = torch.empty_like(full["phi"]); p2 = torch.empty_like(full["p"])
    full["batch"].fit(full["test"], full["ref"], phi2, p2)
    torch.cuda.synchronize()
    assert torch.equal(phi2, full["phi"]) and torch.equal(p2, full["p"])


def test_call_table_is_the_run_length_encoding_of_the_path(full):
    path, calls = full["path"], full["calls"]
    assert full["batch"].n_gsl_errors() == 0
    # every call covers a run of its own type that ends where the state changes
    idx = np.random.default_rng(0).choice(len(calls), size=min(4000, len(calls)), replace=False)
    for c in calls[idx]:
        s, a, b = int(c["sample"]), int(c["start_exon"]), int(c["end_exon"])
        col = path[a:b + 1, s]
        assert col[-1] == c["type"] and np.all(col != 0)
        run = 1
        while b - run >= a and path[b - run, s] == c["type"]:
            run += 1
        assert run == c["nexons"]
        lo, hi = full["chrom_off"][c["chrom"]], full["chrom_off"][c["chrom"] + 1]
        assert lo <= a and b < hi
        assert b + 1 == hi or path[b + 1, s] != c["type"]
        assert a == lo or path[a - 1, s] == 0
    # the number of calls equals the number of non-zero runs (chromosome ends close a run)
    nz = path != 0
    ends = nz.copy()
    ends[:-1] &= (path[:-1] != path[1:])
    last = np.asarray(full["chrom_off"][1:]) - 1
    ends[last] = nz[last]
    assert int(ends.sum()) == len(calls)
    # planted CNVs are found: the overwhelming majority of exons are called normal
    assert 0.0005 < nz.mean() < 0.02


def test_fused_mode_gives_identical_results_at_full_size(full):
    """emissions + Viterbi as ONE kernel, likelihood matrix not materialised: same path, same calls."""
    b = full["edlib"].Batch(full["plan"], S)
    b.set_fused(True); b.keep_loglik(False)
    b.run(full["test"], full["ref"], full["phi"], full["p"])
    assert np.array_equal(b.path(), full["path"])
    assert np.array_equal(b.calls(), full["calls"])
    assert np.array_equal(b.call_info(), full["batch"].call_info())
    b.close()


def test_subset_of_samples_gives_the_same_columns(full):
    torch = full["torch"]
    cols = slice(192, 256)
    b = full["edlib"].Batch(full["plan"], 64)
    b.run(full["test"][:, cols].contiguous(), full["ref"][:, cols].contiguous(), full["phi"][cols].contiguous(),
          full["p"][cols].contiguous())
    assert np.array_equal(b.path(), full["path"][:, cols])
    sub = full["calls"][(full["calls"]["sample"] >= 192) & (full["calls"]["sample"] < 256)].copy()
    sub["sample"] -= 192
    assert np.array_equal(b.calls(), sub)
    b.close()
    torch.cuda.synchronize()


# ---- BASELINE.json configs[3]: the whole 200 000 x 8192 cohort on ONE GPU (1.6e9 cells, ~70 GB of HBM, cell indices
# ---- beyond 2^31) -- what 8 GPUs each do an eighth of
def test_config3_whole_cohort_200k_x_8192_on_one_gpu(edlib, oracle):
    torch = pytest.importorskip("torch")
    from exomedepth_amd import synth
    S8 = 8192
    dev = torch.device("cuda", 0)
    chrom_off, start, end = synth.exon_design(E, C, seed=20250620)
    test, ref, p, phi = synth.counts_torch(chrom_off, S8, dev, seed=20250699)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S8)
    try:
        phi_f = torch.empty(S8, dtype=torch.float64, device=dev)
        p_f = torch.empty(S8, dtype=torch.float64, device=dev)
        batch.fit(test, ref, phi_f, p_f)
        batch.run(test, ref, phi_f, p_f)
        calls = batch.calls()
        assert batch.n_gsl_errors() == 0
        # the whole likelihood matrix (39 GB) against the per-cell evaluation, on the device
        ncmp, nbad, first = batch.verify_emissions(test, ref, phi_f, p_f)
        assert (ncmp, nbad) == (3 * E * S8, 0), first
        path = batch.path()
        phi_h, p_h = phi_f.cpu().numpy(), p_f.cpu().numpy()
        for s in (0, 4095, 4096, 8191):                                  # whole columns against the checker
            t = test[:, s].cpu().numpy(); r = ref[:, s].cpu().numpy()
            ophi, op, _, _ = oracle.fit_mle_hist(t, r)
            assert abs(phi_h[s] - ophi) < 1e-7 * ophi and abs(p_h[s] - op) < 1e-8 * op, ("fit", s)
            ell, nerr = oracle.get_loglike_matrix(phi_h[s], p_h[s], t + r, t, 1.0, oracle.PORTABLE)
            epath, ecalls = oracle.callcnvs(ell, chrom_off, start, end)
            assert nerr == 0 and np.array_equal(path[:, s].astype(np.int8), epath), ("path", s)
            mine = calls[calls["sample"] == s]
            assert len(mine) == len(ecalls)
            for k, name in enumerate(("start_exon", "end_exon", "type", "nexons")):
                assert np.array_equal(mine[name] + (1 if k < 2 else 0), ecalls[:, k].astype(np.int64)), (name, s)
        # properties over all 1.6e9 cells: the call table is the run-length encoding of the path ...
        nz = path != 0
        ends = nz.copy()
        ends[:-1] &= (path[:-1] != path[1:])
        last = np.asarray(chrom_off[1:]) - 1
        ends[last] = nz[last]
        assert int(ends.sum()) == len(calls)
        assert np.array_equal(np.bincount(calls["sample"], minlength=S8), ends.sum(axis=0))
        # ... and a slab of columns run as its own batch gives the same columns (independence of samples)
        cols = slice(5120, 5184)
        b = edlib.Batch(plan, 64)
        b.run(test[:, cols].contiguous(), ref[:, cols].contiguous(), phi_f[cols].contiguous(), p_f[cols].contiguous())
        assert np.array_equal(b.path(), path[:, cols])
        b.close()
    finally:
        batch.close(); plan.close()
        del test, ref
        torch.cuda.empty_cache()


def test_config3_whole_cohort_in_the_table_driven_mode(edlib, oracle):
    """BASELINE configs[3]'s whole cohort (200 000 x 8 192) on one GPU in emit mode 2, counts sample-major: every likelihood value within
    the tolerance of the device's per-cell evaluation of the reference's loop, whole columns against the checker's libm flavour (0 discordant
    states / call rows), the call table as the run-length encoding of all 1.6e9 states"""
    torch = pytest.importorskip("torch")
    from exomedepth_amd import synth
    S8 = 8192
    dev = torch.device("cuda", 0)
    chrom_off, start, end = synth.exon_design(E, C, seed=20250620)
    test, ref, p, phi = synth.counts_torch(chrom_off, S8, dev, seed=20250699)
    test, ref = test.t().contiguous(), ref.t().contiguous()           # [S][E]: R's column-major matrix
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S8)
    batch.set_emit_mode(2); batch.set_counts_layout(1)
    try:
        phi_f = torch.empty(S8, dtype=torch.float64, device=dev)
        p_f = torch.empty(S8, dtype=torch.float64, device=dev)
        batch.fit(test, ref, phi_f, p_f)
        batch.run(test, ref, phi_f, p_f)
        assert batch.fit_unconverged()[0] == 0 and batch.n_gsl_errors() == 0
        chk = batch.verify_emissions_tol(test, ref, phi_f, p_f, rel_tol=1e-10, abs_tol=0.0)
        assert chk["compared"] == 3 * E * S8 and chk["beyond"] == 0, chk
        assert chk["max_rel"] < 1e-12
        calls, path = batch.calls(), batch.path()
        phi_h, p_h = phi_f.cpu().numpy(), p_f.cpu().numpy()
        for s in (0, 4096, 8191):
            t = test[s].cpu().numpy(); r = ref[s].cpu().numpy()
            ell, _ = oracle.get_loglike_matrix(phi_h[s], p_h[s], t + r, t, 1.0, oracle.LIBM)
            epath, ecalls = oracle.callcnvs(ell, chrom_off, start, end)
            assert np.array_equal(path[:, s].astype(np.int8), epath), ("path", s)
            mine = calls[calls["sample"] == s]
            assert len(mine) == len(ecalls)
            for k, name in enumerate(("start_exon", "end_exon", "type", "nexons")):
                assert np.array_equal(mine[name] + (1 if k < 2 else 0), ecalls[:, k].astype(np.int64)), (name, s)
        nz = path != 0
        ends = nz.copy()
        ends[:-1] &= (path[:-1] != path[1:])
        last = np.asarray(chrom_off[1:]) - 1
        ends[last] = nz[last]
        assert int(ends.sum()) == len(calls)
    finally:
        batch.close(); plan.close()
