"""select.reference.set for every sample of a cohort at once (ed_cohort_select_reference_sets: one binary64 Gram matrix on the
matrix cores for the S x S correlations, the K x S cumulative references fitted as one batch, the aggregate reference of every sample
on the device) against S independent calls of the single-test entry -- the loop of reference vignette/vignette.Rnw:390-402 over
R/optimize_reference_set.R:53-148."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cohort(E=20000, S=64, seed=7, depth=90.0):
    rng = np.random.default_rng(seed)
    lam = rng.lognormal(np.log(depth), 0.7, E)
    sf = rng.lognormal(0, 0.25, S)
    # samples come in batches of correlated noise (what makes some references better than others)
    grp = rng.integers(0, 6, S)
    gnoise = rng.normal(0, 0.12, (E, 6))
    own = rng.normal(0, 0.05, (E, S))
    mu = lam[:, None] * sf[None, :] * np.exp(gnoise[:, grp] + own)
    return rng.poisson(mu).astype(np.int32), rng.integers(80, 600, E).astype(float)


@pytest.mark.parametrize("nred", [0, 5000])
def test_cohort_selection_equals_one_call_per_sample(edlib, nred):
    counts, bl = _cohort()
    E, S = counts.shape
    res = edlib.cohort_select_reference_sets(counts, bl, nred, max_refs=32, want_correlations=True)
    K = res["choice"].shape[1]
    agg = res["reference"].to_host().reshape(E, S)
    n_same = 0
    for t in range(S):
        others = np.delete(np.arange(S), t)
        one = edlib.select_reference_set(counts[:, t], np.ascontiguousarray(counts[:, others]), bl, nred)
        st = one["summary.stats"]
        want_choice = [int(others[int(n[1:]) - 1]) for n in one["reference.choice"]]
        got_choice = [int(v) for v in res["choice"][t, :res["n_chosen"][t]]]
        assert got_choice == want_choice, (t, got_choice, want_choice)
        assert res["n.bins"] == one["n.bins"]
        rows = res["summary.stats"][t]
        k = min(K, len(st))
        assert np.array_equal(rows["ref_index"][:k], others[st["ref_index"][:k]])            # the same order of candidates
        assert np.allclose(rows["correlation"][:k], st["correlation"][:k], rtol=0, atol=1e-12)
        # statistics of the cumulative references the R loop reaches (NaN beyond its early exit on both sides)
        for f, tol in (("phi", 1e-7), ("mean_p", 1e-8), ("median_depth", 0), ("ratio_sd", 1e-8), ("expected_BF", 1e-7)):     # (the fits are held to 1e-8: the row-major passes of the single-test entry stop at a step of 2e-5, the cohort entry's columns at 1e-7)
            a, b = rows[f][:k], st[f][:k]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (t, f)
            m = ~np.isnan(a)
            assert np.allclose(a[m], b[m], rtol=tol, atol=0), (t, f, np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))
        assert np.array_equal(rows["selected"][:k], st["selected"][:k])
        assert np.array_equal(agg[:, t], counts[:, want_choice].sum(axis=1))               # the aggregate reference, every exon
        n_same += 1
    assert n_same == S
    c = res["correlations"]
    assert np.allclose(np.diag(c), 1.0, atol=1e-12) and np.allclose(c, c.T, atol=1e-14)
    z = counts / (bl[:, None] * counts.sum(axis=0)[None, :] / 1e6)   # all bins: only a sanity check of the scale
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1] - c[0, 1]) < 0.1


def _same_selection(a, b, stats_tol):
    assert np.array_equal(a["n_chosen"], b["n_chosen"]) and np.array_equal(a["choice"], b["choice"]) and a["n.bins"] == b["n.bins"]
    ra, rb = a["summary.stats"], b["summary.stats"]
    assert np.array_equal(ra["ref_index"], rb["ref_index"]) and np.array_equal(ra["selected"], rb["selected"])
    reached = ~np.isnan(ra["expected_BF"])
    assert np.array_equal(reached, ~np.isnan(rb["expected_BF"]))
    assert np.array_equal(ra["median_depth"][reached], rb["median_depth"][reached])          # order statistics: exact
    # (the row-major passes stop at a step of 2e-5, which leaves ~C 4e-10 with C up to a few tens in the dispersion: tests/test_gpu_fit.py)
    for f, tol in (("phi", 10 * stats_tol), ("mean_p", stats_tol), ("ratio_sd", stats_tol), ("expected_BF", 10 * stats_tol)):
        x, y = ra[f][reached], rb[f][reached]
        assert np.allclose(x, y, rtol=tol, atol=0), (f, float(np.max(np.abs(x - y) / np.abs(y))))


@pytest.mark.parametrize("depth", [90.0, 140.0, 250.0, 450.0, 4000.0])
def test_column_major_and_row_major_forms_agree(edlib, monkeypatch, depth):
    """The chunk loop has two forms (csrc/edrefcohort.inc): one workgroup per cumulative reference on the column's count histograms (k_rc_column:
    the fit on tail counts, the median and RatioSd from the same bins; two geometries, chosen per column from its mean depth, the small one handing
    on what it cannot hold), and the row-major kernels of rounds 2-5.  Same choices, the same medians, the other statistics to the fits' rounding --
    at depths where the small bins hold every column (with and without values beyond them, which are kept as sorted values), where the deep
    prefixes take the large bins, and where the chunk is left to the row-major kernels; and with every column forced through one geometry first."""
    counts, bl = _cohort(E=9000, S=48, seed=21, depth=depth)
    got = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=32, want_reference=False)
    path = edlib.refcohort_last_path()
    by_columns = depth < 4000.0
    assert (path["chunks_by_columns"], path["chunks_row_major"]) == ((1, 0) if by_columns else (0, 1)), path
    if depth == 90.0:
        assert path["columns_beyond_bins"] == 0 and path["columns_large_geometry"] == 0, path
    if depth >= 250.0 and by_columns:
        assert path["columns_large_geometry"] > 0, path
    if by_columns:
        assert 0 < path["max_newton_iterations"] < 30
    monkeypatch.setenv("ED_REFCOHORT_ROWMAJOR", "1")
    ref = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=32, want_reference=False)
    assert edlib.refcohort_last_path()["chunks_by_columns"] == 0
    monkeypatch.delenv("ED_REFCOHORT_ROWMAJOR")
    _same_selection(got, ref, 3e-8)
    again = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=32, want_reference=False)       # the same bits run to run
    for f in ("phi", "mean_p", "ratio_sd", "expected_BF", "median_depth"):
        assert np.array_equal(got["summary.stats"][f], again["summary.stats"][f], equal_nan=True), f
    if depth in (140.0, 250.0, 4000.0):   # every column through the small geometry first: what it cannot hold is handed on (and at 4 000 the large one gives up too)
        monkeypatch.setenv("ED_REFCOHORT_GEOMETRY", "1")
        forced = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=32, want_reference=False)
        p2 = edlib.refcohort_last_path()
        monkeypatch.delenv("ED_REFCOHORT_GEOMETRY")
        if depth == 140.0:
            assert p2["chunks_by_columns"] == 1 and p2["columns_beyond_bins"] > 0, p2      # values beyond the small bins, kept in the lists
        if depth == 250.0:
            assert p2["chunks_by_columns"] == 1 and p2["columns_large_geometry"] > 0, p2
        if depth == 4000.0:
            assert p2["chunks_row_major"] == 1, p2
        _same_selection(forced, ref, 3e-8)
    if depth == 90.0:                     # the large geometry on shallow data: the same fits from other bins
        monkeypatch.setenv("ED_REFCOHORT_GEOMETRY", "2")
        forced = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=32, want_reference=False)
        p2 = edlib.refcohort_last_path()
        assert p2["columns_large_geometry"] > 1000 and p2["chunks_by_columns"] == 1, p2
        monkeypatch.delenv("ED_REFCOHORT_GEOMETRY")
        _same_selection(forced, got, 1e-11)


@pytest.mark.parametrize("depth", [90.0, 250.0])
def test_column_form_against_the_long_double_mle(edlib, depth):
    """The cohort entry's statistics against the CPU restatement of R/optimize_reference_set.R:53-148 whose fits are the long-double MLE on
    sufficient statistics (oracle/refset_oracle.py: select_reference_set_lean): the column form iterates to a step of 1e-7 on sums of positive
    terms and sits 1e-11 from it in phi, 1e-14 in the mean, 1e-12 in RatioSd and the expected Bayes factor (the row-major passes, which stop at a
    step of 2e-5: 3e-9, 3e-11, 6e-10, 1e-9 on the same columns).  Both geometries."""
    from oracle import refset_oracle as ro
    counts, bl = _cohort(E=9000, S=48, seed=21, depth=depth)
    res = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=32, want_reference=False)
    assert (edlib.refcohort_last_path()["columns_large_geometry"] > 0) == (depth == 250.0)
    for t in (0, 17, 40):
        others = np.delete(np.arange(48), t)
        one = ro.select_reference_set_lean(counts[:, t], np.ascontiguousarray(counts[:, others]), bl, 0)
        rows = res["summary.stats"][t]
        k = len(rows)
        assert np.array_equal(rows["ref_index"], others[one["order"][:k]])
        assert res["n_chosen"][t] == one["n_chosen"]
        for f, g, tol in (("phi", "phi", 1e-10), ("mean_p", "mean_p", 1e-12), ("ratio_sd", "RatioSd", 2e-11), ("expected_BF", "expected_BF", 5e-11), ("median_depth", "median_depth", 0)):
            a, b = rows[f], one[g][:k]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (t, f)
            m = ~np.isnan(a)
            assert np.allclose(a[m], b[m], rtol=tol, atol=0), (t, f, float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))))


def test_a_negative_count_leaves_the_chunk_to_the_row_major_kernels(edlib, monkeypatch):
    """not a count: the column form does not interpret it (its bins are indexed by the counts) -- the columns that hold one raise the flag, and the
    chunk gets what the row-major kernels' per-cell arithmetic makes of it, as before"""
    counts, bl = _cohort(E=6000, S=24, seed=5)
    counts[counts.sum(axis=1).argsort()[3000], 7] = -1           # in a bin the selection keeps (total near the median)
    got = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=23, want_reference=False)
    path = edlib.refcohort_last_path()
    assert path["chunks_row_major"] == 1 and path["chunks_by_columns"] == 0, path
    monkeypatch.setenv("ED_REFCOHORT_ROWMAJOR", "1")
    ref = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=23, want_reference=False)
    monkeypatch.delenv("ED_REFCOHORT_ROWMAJOR")
    assert np.array_equal(got["choice"], ref["choice"])
    for f in ("phi", "mean_p", "ratio_sd", "expected_BF", "median_depth"):
        assert np.array_equal(got["summary.stats"][f], ref["summary.stats"][f], equal_nan=True), f


def test_a_hand_over_the_host_did_not_predict(edlib, monkeypatch):
    """The host gives a column its geometry from its MEAN depth; a cohort whose bins come in two depths (most shallow, one in twenty 25 times deeper -- still
    below the 90 % quantile of the totals, so selected) has small means and thousands of cells beyond the small bins: the small geometry's workgroups hand
    such columns on, and the large launch, which was not issued beside them, is issued after all.  Same answer as the row-major kernels."""
    rng = np.random.default_rng(3)
    E, S = 40000, 32
    deep = rng.random(E) < 0.14
    lam = np.where(deep, 1000.0, 40.0) * rng.lognormal(0, 0.15, E)
    sf = rng.lognormal(0, 0.2, S)
    counts = rng.poisson(lam[:, None] * sf[None, :] * np.exp(rng.normal(0, 0.08, (E, S)))).astype(np.int32)
    bl = rng.integers(80, 600, E).astype(float)
    got = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=31, want_reference=False)
    path = edlib.refcohort_last_path()
    assert path["chunks_by_columns"] == 1 and path["columns_large_geometry"] > 0, path
    monkeypatch.setenv("ED_REFCOHORT_ROWMAJOR", "1")
    ref = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=31, want_reference=False)
    monkeypatch.delenv("ED_REFCOHORT_ROWMAJOR")
    _same_selection(got, ref, 3e-8)


def test_more_bins_than_the_column_form_takes(edlib):
    """more than 65 535 selected bins: 16-bit tail counts do not hold them -- the row-major kernels, as before"""
    counts, bl = _cohort(E=110_000, S=12, seed=4, depth=60.0)
    res = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=11, want_reference=False)
    assert res["n.bins"] > 65535
    path = edlib.refcohort_last_path()
    assert path["chunks_by_columns"] == 0 and path["chunks_row_major"] >= 1
    t = 5
    others = np.delete(np.arange(12), t)
    one = edlib.select_reference_set(counts[:, t], np.ascontiguousarray(counts[:, others]), bl, 0)
    assert [int(v) for v in res["choice"][t, :res["n_chosen"][t]]] == [int(others[int(nm[1:]) - 1]) for nm in one["reference.choice"]]


def test_a_short_max_refs_falls_back_instead_of_changing_the_answer(edlib):
    """with K = 3 the R loop (which stops at the first i > 2 with mean.p < 0.05) cannot finish inside the K prefixes: every test goes
    through the single-test entry, and the choices are those of the K = 32 run -- or the call says that a choice does not fit"""
    counts, bl = _cohort(E=8000, S=24, seed=3)
    full = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=23, want_reference=False)
    if full["n_chosen"].max() <= 3:
        short = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=3, want_reference=False)
        assert np.array_equal(short["n_chosen"], full["n_chosen"])
        for t in range(24):
            assert np.array_equal(short["choice"][t, :short["n_chosen"][t]], full["choice"][t, :full["n_chosen"][t]])
    else:
        with pytest.raises(edlib.EdError, match="larger max_refs"):
            edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=3, want_reference=False, grow_max_refs=False)
        # ... and the wrapper's default does what the message says: max_refs doubled until every choice fits, the same answer
        grown = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=3, want_reference=False)
        assert grown["choice"].shape[1] > 3 and np.array_equal(grown["n_chosen"], full["n_chosen"])
        for t in range(24):
            assert np.array_equal(grown["choice"][t, :grown["n_chosen"][t]], full["choice"][t, :full["n_chosen"][t]])


def test_cohort_selection_feeds_the_cohort_pipeline(edlib):
    """counts -> reference sets -> aggregate reference (device) -> fit + emissions + Viterbi for the whole cohort: the canonical
    workflow end to end on the device, equal to the batch interface fed the host-built aggregate reference"""
    from exomedepth_amd import synth
    counts, bl = _cohort(E=12000, S=48, seed=9)
    E, S = counts.shape
    chrom_off, start, end = synth.exon_design(E, 4, seed=2)
    res = edlib.cohort_select_reference_sets(counts, (end - start) / 1000.0, 5000)
    plan = edlib.Plan(chrom_off, start, end)
    co = edlib.Cohort(plan, S, 2)
    dcounts = edlib.DeviceArray(counts)
    t = co.submit(dcounts, res["reference"], n_samples=S)
    got = co.results(t, S, path=True)
    ref_h = np.stack([counts[:, res["choice"][s, :res["n_chosen"][s]]].sum(axis=1) for s in range(S)], axis=1).astype(np.int32)
    b = edlib.Batch(plan, S)
    dphi, dexp = edlib.DeviceArray(np.zeros(S)), edlib.DeviceArray(np.zeros(S))
    b.fit(counts, ref_h, dphi, dexp)
    b.run(counts, ref_h, dphi, dexp)
    assert got["calls"].tobytes() == b.calls().tobytes() and got["path"].tobytes() == b.path().tobytes()
    assert got["phi"].tobytes() == dphi.to_host().tobytes()
    b.close(); co.close(); plan.close()


def test_config4_geometry_every_sample_against_all_others_500k_x_2048(edlib):
    """BASELINE configs[4]'s geometry at full size -- 500 000 bins x 2048 samples, every sample in turn the test and the other 2047 its
    candidates (n.bins.reduced = 10 000, as the reference's vignette uses): one call; spot columns against the single-test entry."""
    torch = pytest.importorskip("torch")
    import time
    Eb, S = 500_000, 2048
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(11)
    lam = torch.empty(Eb, device=dev, dtype=torch.float32).log_normal_(float(np.log(60.0)), 0.7, generator=g)
    sig = torch.linspace(0.03, 0.3, S, device=dev)[torch.randperm(S, device=dev, generator=g)]
    counts = torch.empty((Eb, S), device=dev, dtype=torch.int32)
    for lo in range(0, Eb, 16384):
        hi = min(lo + 16384, Eb)
        noise = torch.exp(torch.randn((hi - lo, S), device=dev, generator=g) * sig[None, :])
        counts[lo:hi] = torch.poisson(lam[lo:hi, None] * noise, generator=g).to(torch.int32)
    length = torch.randint(60, 600, (Eb,), device=dev, generator=g).cpu().numpy().astype(np.float64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = edlib.cohort_select_reference_sets(counts, length, 10000, max_refs=32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("cohort select.reference.set, 500 000 x 2048, n.bins.reduced 10 000: %.3f s, %.1f references chosen on average" % (dt, res["n_chosen"].mean()))
    assert res["n.bins"] in (10000, 10001) and np.all(res["n_chosen"] >= 1) and np.all(res["n_chosen"] <= 32)
    ref = res["reference"]
    from exomedepth_amd import dist as eddist
    agg = torch.as_tensor(eddist._DevicePointer(ref.ptr.value, (Eb, S), "<i4"), device=dev)
    for t in (0, 777, 2047):
        others = torch.cat([counts[:, :t], counts[:, t + 1:]], dim=1).contiguous()
        one = edlib.select_reference_set(counts[:, t].contiguous(), others, length, 10000)
        want = [i + (1 if i >= t else 0) for i in [int(n[1:]) - 1 for n in one["reference.choice"]]]
        assert [int(v) for v in res["choice"][t, :res["n_chosen"][t]]] == want
        assert torch.equal(agg[:, t], counts[:, want].sum(dim=1).to(torch.int32))
        del others


def test_one_rank_of_eight_200k_x_8192(edlib):
    """What every rank of `bench.py --gpus 8` runs in its workflow leg after the all-gather of the count slabs (BASELINE configs[3]'s cohort): 200 000 bins x
    8 192 samples on the device, the rank's own 1 024 samples as tests, all 8 192 as candidates.  The last rank's share; spot tests against the single-test
    entry (same columns, same order) and the aggregate reference against the column sums."""
    torch = pytest.importorskip("torch")
    from exomedepth_amd import synth, dist as eddist
    E, S, W, rank = 200_000, 1024, 8, 7
    dev = torch.device("cuda", 0)
    torch.manual_seed(20250623)            # (synth draws its beta variates from torch's global generator)
    chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
    counts = torch.cat([synth.counts_torch(chrom_off, S, dev, seed=20250623 + r, mean_depth=100.0)[0] for r in range(W)], dim=1).contiguous()
    bl = (np.asarray(end) - np.asarray(start)) / 1000.0
    res = edlib.cohort_select_reference_sets(counts, bl, 10000, max_refs=32, test_range=(rank * S, (rank + 1) * S))
    # (with 8 191 candidates a choice of 33 - 35 references does occur: the wrapper then doubles max_refs, as the C entry's message asks)
    assert res["n.bins"] in (10000, 10001) and np.all(res["n_chosen"] >= 1) and np.all(res["n_chosen"] <= res["choice"].shape[1])
    agg = torch.as_tensor(eddist._DevicePointer(res["reference"].ptr.value, (E, S), "<i4"), device=dev)
    for t in (rank * S + 517, rank * S + int(np.argmax(res["n_chosen"])), (rank + 1) * S - 1):
        keep = [c for c in range(S * W) if c != t]
        one = edlib.select_reference_set(counts[:, t].contiguous(), counts[:, keep].contiguous(), bl, 10000, names=[str(c) for c in keep])
        want = [int(c) for c in one["reference.choice"]]
        tl = t - rank * S
        assert [int(v) for v in res["choice"][tl, :res["n_chosen"][tl]]] == want
        assert torch.equal(agg[:, tl], counts[:, want].sum(dim=1).to(torch.int32))


def test_test_ranges_are_the_rows_of_the_whole_cohort_call(edlib):
    """ed_cohort_select_reference_sets_range: the tests [t0, t1) against all S candidates -- one rank's share of a sample-sharded cohort
    (vignette/vignette.Rnw:390-402 for its own samples) -- returns exactly the corresponding rows / columns of the whole-cohort call:
    choices, summary statistics, correlations, aggregate references"""
    from exomedepth_amd import synth
    E, S = 6000, 100
    chrom_off, start, end = synth.exon_design(E, 4, 31)
    counts, _, _, _, _ = synth.counts_numpy(chrom_off, S, 31, n_segments=2, mean_depth=80.0)
    bl = (end - start) / 1000.0
    whole = edlib.cohort_select_reference_sets(counts, bl, 2000, want_correlations=True)
    ref_whole = whole["reference"].to_host()
    for t0, t1 in ((0, 37), (37, 100), (50, 51), (31, 97)):       # block boundaries of the Gram tiles (32) inside and outside
        part = edlib.cohort_select_reference_sets(counts, bl, 2000, want_correlations=True, test_range=(t0, t1))
        assert part["n.bins"] == whole["n.bins"]
        assert np.array_equal(part["n_chosen"], whole["n_chosen"][t0:t1]) and np.array_equal(part["choice"], whole["choice"][t0:t1])
        assert part["summary.stats"].tobytes() == whole["summary.stats"][t0:t1].tobytes()
        assert np.array_equal(part["correlations"], whole["correlations"][t0:t1])
        assert np.array_equal(part["reference"].to_host(), ref_whole[:, t0:t1])
    with pytest.raises(Exception, match="tests"):
        edlib.cohort_select_reference_sets(counts, bl, 2000, test_range=(5, 5))


@pytest.mark.parametrize("S,E", [(64, 20000), (150, 4001), (1100, 3000), (2100, 700)])
def test_sample_major_outputs_are_the_transposes(edlib, S, E):
    """ed_cohort_select_reference_sets_sm: the aggregate references sample-major (and the count matrix transposed alongside) are, bit for
    bit, the transposes of what ed_cohort_select_reference_sets writes -- tiles of 32 / 16 / 8 rows, ragged last tile, a share of the
    tests -- and feed a cohort with counts_layout = 1 to the same calls"""
    counts, bl = _cohort(E=E, S=S, seed=11 + S)
    a = edlib.cohort_select_reference_sets(counts, bl, 2000, max_refs=32)
    want = a["reference"].to_host().reshape(E, S)
    cs = edlib.DeviceArray(nbytes=E * S * 4)
    cs.host_dtype, cs.shape = np.dtype(np.int32), (S, E)
    b = edlib.cohort_select_reference_sets(counts, bl, 2000, max_refs=32, sample_major=True, counts_sm_out=cs)
    assert np.array_equal(a["choice"], b["choice"]) and np.array_equal(a["n_chosen"], b["n_chosen"])
    assert np.array_equal(b["reference"].to_host().reshape(S, E), want.T)
    assert np.array_equal(cs.to_host().reshape(S, E), counts.T)
    t0, t1 = S // 3, S // 3 + max(1, S // 4)
    c = edlib.cohort_select_reference_sets(counts, bl, 2000, max_refs=32, sample_major=True, test_range=(t0, t1))
    assert np.array_equal(c["reference"].to_host().reshape(t1 - t0, E), want.T[t0:t1])
    # a candidate list longer than the one-pass kernel holds: the same result through the [n_bins][n_tests] form
    d = edlib.cohort_select_reference_sets(counts, bl, 2000, max_refs=min(S - 1, 140), sample_major=True, counts_sm_out=cs)
    assert np.array_equal(d["reference"].to_host().reshape(S, E), want.T) and np.array_equal(cs.to_host().reshape(S, E), counts.T)
