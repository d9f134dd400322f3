"""select.reference.set on the GPU against the CPU restatement of the R code (oracle/refset_oracle.py).
Parity status: unpinned against the reference (the R code needs R, aod and VGAM); tolerance-level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cohort(E, R, seed, depth=60.0, noise=(0.02, 0.35)):
    """A test sample plus R candidate references of graded similarity: reference r carries a multiplicative
    per-exon distortion whose size grows with r, so correlations are well separated."""
    rng = np.random.default_rng(seed)
    lam = rng.lognormal(np.log(depth), 0.7, E)
    length = rng.integers(60, 600, E).astype(np.float64)
    test = rng.poisson(lam * 1.0)
    sig = np.linspace(noise[0], noise[1], R)[rng.permutation(R)]
    refs = np.stack([rng.poisson(lam * rng.lognormal(0.0, s, E) * rng.uniform(0.7, 1.3)) for s in sig], axis=1)
    return test.astype(np.int32), refs.astype(np.int32), length


@pytest.mark.parametrize("E,R,reduced,use_len", [(6000, 10, 0, True), (9000, 7, 2500, False)])
def test_select_reference_set_matches_the_restated_r_code(edlib, oracle, E, R, reduced, use_len):
    from oracle import refset_oracle as ro
    test, refs, length = _cohort(E, R, seed=100 + R)
    bl = length if use_len else None
    got = edlib.select_reference_set(test, refs, bin_length=bl, n_bins_reduced=reduced)
    exp = ro.select_reference_set(test, refs, bin_length=bl, n_bins_reduced=reduced)
    st = got["summary.stats"]
    assert got["n.bins"] == exp["n_bins"]
    assert np.array_equal(st["ref_index"], exp["order"])
    assert np.allclose(st["correlation"], exp["correlations"], rtol=0, atol=1e-12)
    for mine, theirs, tol in (("phi", "phi", 1e-7), ("mean_p", "mean_p", 1e-8), ("median_depth", "median_depth", 0.0),
                              ("ratio_sd", "RatioSd", 1e-8), ("expected_BF", "expected_BF", 1e-7)):
        a, b = st[mine], exp[theirs]
        assert np.array_equal(np.isnan(a), np.isnan(b)), mine
        m = ~np.isnan(b)
        assert np.all(np.abs(a[m] - b[m]) <= tol * np.abs(b[m])), (mine, a, b)
    assert len(got["reference.choice"]) == exp["n_chosen"]
    assert int(np.argmax(st["selected"])) == exp["n_chosen"] - 1 and st["selected"].sum() == 1
    # the best-correlated references are the least distorted ones, and adding the worst ones does not pay
    assert 1 <= exp["n_chosen"] <= R


def test_select_reference_set_early_exit_and_guards(edlib, oracle):
    from oracle import refset_oracle as ro
    # many deep references: the test proportion drops below 0.05 and the R loop breaks (:130)
    test, refs, _ = _cohort(5000, 30, seed=7, depth=40.0, noise=(0.05, 0.1))
    got = edlib.select_reference_set(test, refs)
    exp = ro.select_reference_set(test, refs)
    st = got["summary.stats"]
    assert np.array_equal(np.isnan(st["expected_BF"]), np.isnan(exp["expected_BF"]))
    assert np.isnan(st["expected_BF"]).any() and np.array_equal(np.isnan(st["phi"]), np.isnan(exp["phi"]))
    assert len(got["reference.choice"]) == exp["n_chosen"]
    # coverage guard (:57-61): fewer than 5 bins with more than 2 reads -> first reference
    got = edlib.select_reference_set(np.zeros(100, np.int32), refs[:100])
    assert got["reference.choice"] == ["X1"]
    with pytest.raises(ValueError, match="bin.length contains 1 zero"):
        edlib.select_reference_set(test, refs, bin_length=np.r_[0.0, np.ones(4999)])


def test_get_power_betabinom_known_properties(oracle):
    from oracle import refset_oracle as ro
    # reference R/tools.R:123-125 examples: positive when the alternative differs, exactly 0 when it does not
    assert ro.get_power_betabinom(200, 0.1, 0.2, 0.6) > 1.0
    assert abs(ro.get_power_betabinom(200, 0.1, 0.2, 0.2)) < 1e-12


def test_prefix_shares_merge_to_the_whole(edlib):
    """The multi-GPU decomposition (dist.select_reference_set_sharded): shares of the sorted prefix axis computed
    independently, merged and finalised, are the single-GPU result bit for bit -- including a case where the
    loop's early exit (mean.p < 0.05) falls inside the second share."""
    for seed, R, scale in ((7, 9, 1.0), (8, 12, 40.0)):
        test, refs, length = _cohort(5000, R, seed=seed)
        refs = (refs * scale).astype(np.int32)          # scale 40: the cumulative reference dwarfs the test -> mean.p < 0.05
        whole = edlib.select_reference_set(test, refs, bin_length=length)
        cut = R // 2
        a = edlib.select_reference_set(test, refs, bin_length=length, prefix_window=(0, cut))
        b = edlib.select_reference_set(test, refs, bin_length=length, prefix_window=(cut, R))
        assert a["reference.choice"] is None and a["n.bins"] == whole["n.bins"]
        merged = a["summary.stats"].copy()
        merged[cut:] = b["summary.stats"][cut:]
        assert np.array_equal(a["summary.stats"]["ref_index"], b["summary.stats"]["ref_index"])
        fin = edlib.refset_finalize(merged)
        w = whole["summary.stats"]
        for name in w.dtype.names:
            x, y = fin["summary.stats"][name], w[name]
            assert np.array_equal(x.view(np.uint8 if x.dtype.itemsize == 1 else ("u4" if x.dtype.itemsize == 4 else "u8")),
                                  y.view("u4" if y.dtype.itemsize == 4 else "u8")), name
        assert fin["reference.choice"] == whole["reference.choice"]
        if scale > 1:
            assert np.isnan(w["expected_BF"]).any()      # the early exit did trigger



def test_get_power_betabinom_walk_against_mpmath(edlib):
    """The default case walks the two mass functions (k_rs_power_walk, round 6: step ratios, one logarithm per x) instead of three lnbeta per
    term.  Against 50-digit arithmetic at every tile boundary of the walk (256 values per tile), on a near-null alternative (where the sum is
    a cancellation of terms of both signs: the term-by-term form keeps 1e-8 there, the walk 1e-10) and against the term-by-term checker on a
    spread of parameters; as phi -> 0 the walk approaches the binomial case (where differences of lnbeta values of order 1 / phi have no digits left)."""
    import mpmath as mp
    from oracle import refset_oracle as ro
    mp.mp.dps = 40

    def exact(size, phi, p, alt):
        phi, p, alt = mp.mpf(float(phi)), mp.mpf(float(p)), mp.mpf(float(alt))
        a, b, aa, ab = p * (1 - phi) / phi, (1 - p) * (1 - phi) / phi, alt * (1 - phi) / phi, (1 - alt) * (1 - phi) / phi
        lb = lambda x, y: mp.loggamma(x) + mp.loggamma(y) - mp.loggamma(x + y)
        tot = mp.mpf(0)
        for x in range(int(size) + 1):
            la = lb(aa + x, ab + size - x) - lb(aa, ab)
            tot += mp.e ** (-mp.log(size + 1) - lb(size - x + 1, x + 1) + la) * (la - (lb(a + x, b + size - x) - lb(a, b)))
        return float(tot * mp.log10(mp.e))

    cases = [(0, .05, .3, .2), (1, .05, .3, .2), (3, .2, .5, .4), (4, .2, .5, .4), (5, .2, .5, .4), (255, .01, .1, .0526), (256, .01, .1, .0526),
             (257, .01, .1, .0526), (511, .003, .11, .06), (512, .003, .11, .06), (1025, .02, .4, .25), (2100, 1e-6, .02, .0101),
             (1500, 0.037713, 0.294922, 0.297696), (700, .5, .9, .05)]
    size, phi, p, alt = (np.array(c, dtype=float) for c in zip(*cases))
    got = edlib.get_power_betabinom(size, phi, p, alt)
    want = np.array([exact(*c) for c in cases])
    assert np.allclose(got, want, rtol=2e-10, atol=1e-15), (got, want)
    rng = np.random.default_rng(3)
    n = 300
    size = rng.integers(0, 6000, n).astype(float)
    phi = 10.0 ** rng.uniform(-6, -0.3, n)
    p = rng.uniform(0.01, 0.9, n)
    alt = np.where(rng.random(n) < 0.5, (p / (1 - p) * 0.5) / (1 + p / (1 - p) * 0.5), rng.uniform(0.01, 0.9, n))
    got = edlib.get_power_betabinom(size, phi, p, alt)
    exp = np.array([ro.get_power_betabinom(int(s), f, q, a) for s, f, q, a in zip(size, phi, p, alt)])
    assert np.allclose(got, exp, rtol=1e-8, atol=1e-11)
    tiny = edlib.get_power_betabinom([300.0, 300.0], [1e-70, 1e-9], [.2, .2], [.1, .1])
    binom = float(np.ravel(edlib.get_power_betabinom([300.0], [.5], [.2], [.1], theory=True))[0])
    assert np.isfinite(tiny).all() and abs(tiny[0] - binom) < 1e-6 * binom and abs(tiny[1] - binom) < 1e-5 * binom


def test_get_power_betabinom_standalone(edlib):
    """reference R/tools.R:128-166, default mode, incl. its two documented examples (my.alt.p = my.p gives 0)."""
    from oracle import refset_oracle as ro
    size = np.array([200, 200, 57, 1000, 3000, 0], dtype=float)
    phi = np.array([0.1, 0.1, 0.01, 0.003, 0.02, 0.05])
    p = np.array([0.2, 0.2, 0.1, 0.11, 0.4, 0.3])
    alt = np.array([0.6, 0.2, 0.0526, 0.06, 0.25, 0.2])
    got = edlib.get_power_betabinom(size, phi, p, alt)
    exp = np.array([ro.get_power_betabinom(int(s), f, q, a) for s, f, q, a in zip(size, phi, p, alt)])
    assert abs(got[1]) < 1e-12                                     # identical hypotheses: no evidence expected
    assert got[0] > 1.0 and np.allclose(got, exp, rtol=1e-9, atol=1e-12), (got, exp)
    assert isinstance(edlib.get_power_betabinom(200, 0.1, 0.2, 0.6), float)
    # theory = TRUE: the binomial case (R/tools.R:137-142)
    got = edlib.get_power_betabinom(size[:5], phi[:5], p[:5], alt[:5], theory=True)
    exp = np.array([ro.get_power_binom(int(s), q, a) for s, q, a in zip(size[:5], p[:5], alt[:5])])
    assert abs(got[1]) < 1e-12 and np.allclose(got, exp, rtol=1e-9, atol=1e-12), (got, exp)
    # a binomial separates the hypotheses better than an over-dispersed model of the same means
    assert np.all(got[[0, 2, 3, 4]] > edlib.get_power_betabinom(size[:5], phi[:5], p[:5], alt[:5])[[0, 2, 3, 4]])
    # limit = TRUE (R/tools.R:145-153): the reference averages log10[dbeta(X/size; alt) / dbeta(X/size; null)] over 2000 draws of
    # rbetabinom.ab(alt); here the expectation that average estimates, against scipy's beta-binomial pmf and beta log-densities --
    # and against a seeded Monte-Carlo run of the reference's recipe (2000 draws: agreement within a few standard errors)
    from scipy import stats
    got = edlib.get_power_betabinom(size[:5], phi[:5], p[:5], alt[:5], limit=True)
    rng = np.random.default_rng(5)
    for k in range(5):
        n_, f_, q_, a_ = int(size[k]), phi[k], p[k], alt[k]
        a0, b0 = q_ * (1 - f_) / f_, (1 - q_) * (1 - f_) / f_
        a1, b1 = a_ * (1 - f_) / f_, (1 - a_) * (1 - f_) / f_
        x = np.arange(1, n_)
        w = stats.betabinom.pmf(x, n_, a1, b1)
        lr = (stats.beta.logpdf(x / n_, a1, b1) - stats.beta.logpdf(x / n_, a0, b0)) * np.log10(np.e)
        want = float(np.sum(w * lr))
        assert abs(got[k] - want) <= 1e-8 * max(abs(want), 1e-6), (k, got[k], want)
        draws = stats.betabinom.rvs(n_, a1, b1, size=2000, random_state=rng) / n_
        draws = draws[(draws > 0) & (draws < 1)]
        mc = (stats.beta.logpdf(draws, a1, b1) - stats.beta.logpdf(draws, a0, b0)) * np.log10(np.e)
        assert abs(mc.mean() - got[k]) < 5 * mc.std() / np.sqrt(len(mc)) + 1e-9, (k, mc.mean(), got[k])
    assert abs(got[1]) < 1e-12


# ---- BASELINE.json configs[4]: select.reference.set, 500 000 bins x 2048 candidate references ----
def test_config4_select_reference_set_500k_x_2048(edlib, oracle):
    """Full size through ed_select_reference_set: against the restated R code (memory-lean form, the checker's long-double
    MLE) on every row the R loop reaches before its early exit, on the RAW statistics of a strided set of deeper
    cumulative references, and the two-share decomposition of the multi-GPU path merging to the whole bit for bit."""
    torch = pytest.importorskip("torch")
    from oracle import refset_oracle as ro
    Eb, R = 500_000, 2048
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    lam = torch.empty(Eb, device=dev, dtype=torch.float32).log_normal_(float(np.log(60.0)), 0.7, generator=g)
    sig = torch.linspace(0.02, 0.4, R, device=dev)[torch.randperm(R, device=dev, generator=g)]
    # the test sample carries its own per-bin distortion (8 %): the dispersion of test vs cumulative reference then stays
    # at exome-like values (phi ~ 1e-3 - 1e-4) however many references are summed, instead of decaying to a binomial
    test = torch.poisson(lam * torch.exp(0.08 * torch.randn(Eb, device=dev, generator=g)), generator=g).to(torch.int32)
    refs = torch.empty((Eb, R), device=dev, dtype=torch.int32)
    for lo in range(0, Eb, 16384):
        hi = min(lo + 16384, Eb)
        noise = torch.exp(torch.randn((hi - lo, R), device=dev, generator=g) * sig[None, :])
        refs[lo:hi] = torch.poisson(lam[lo:hi, None] * noise, generator=g).to(torch.int32)
    length = torch.randint(60, 600, (Eb,), device=dev, generator=g).cpu().numpy().astype(np.float64)
    whole = edlib.select_reference_set(test, refs, bin_length=length)
    st = whole["summary.stats"]
    raw_rows = (0, 1, 2, 7, 19, 20, 21, 63, 257)
    exp = ro.select_reference_set_lean(test.cpu().numpy(), refs.cpu().numpy(), bin_length=length, raw_prefixes=raw_rows)
    assert whole["n.bins"] == exp["n_bins"] and exp["n_bins"] > 300_000
    assert np.array_equal(st["ref_index"], exp["order"])
    assert np.allclose(st["correlation"], exp["correlations"], rtol=0, atol=1e-12)
    reached = ~np.isnan(exp["phi"])
    assert 3 <= reached.sum() < 100                                       # the loop's early exit (:130) triggers: mean.p < 0.05
    tol_phi_rows = np.maximum(1e-7, 1e-13 / np.where(reached, exp["phi"], 1.0) ** 2)   # binary64 resolves phi to ~2e-14 / phi^2
    assert np.all(exp["phi"][reached] > 1e-5)
    for mine, theirs, tol in (("phi", "phi", tol_phi_rows), ("mean_p", "mean_p", 1e-8), ("median_depth", "median_depth", 0.0),
                              ("ratio_sd", "RatioSd", 1e-7), ("expected_BF", "expected_BF", 10 * tol_phi_rows)):
        a, b = st[mine], exp[theirs]
        assert np.array_equal(np.isnan(a), np.isnan(b)), mine
        m = ~np.isnan(b)
        assert np.all(np.abs(a[m] - b[m]) <= (tol[m] if np.ndim(tol) else tol) * np.abs(b[m])), (mine, a[m], b[m])
    assert len(whole["reference.choice"]) == exp["n_chosen"] and int(np.argmax(st["selected"])) == exp["n_chosen"] - 1
    # raw statistics of deeper prefixes + the decomposition over two ranks
    cut = R // 2
    a = edlib.select_reference_set(test, refs, bin_length=length, prefix_window=(0, cut))
    b = edlib.select_reference_set(test, refs, bin_length=length, prefix_window=(cut, R))
    merged = a["summary.stats"].copy()
    merged[cut:] = b["summary.stats"][cut:]
    for i in raw_rows:
        want = exp["raw"][i]
        row = merged[i]
        tol_phi = max(1e-7, 1e-13 / want["phi"] ** 2)                     # binary64 resolves phi to ~2e-14 / phi^2
        assert abs(row["phi"] - want["phi"]) <= tol_phi * want["phi"], (i, row, want)
        assert abs(row["mean_p"] - want["mean_p"]) <= 1e-8 * want["mean_p"] and row["median_depth"] == want["median_depth"], (i, row, want)
        assert abs(row["ratio_sd"] - want["RatioSd"]) <= 1e-7 * want["RatioSd"], (i, row, want)
        assert abs(row["expected_BF"] - want["expected_BF"]) <= max(1e-7, 10 * tol_phi) * abs(want["expected_BF"]), (i, row, want)
    assert np.all(np.isfinite(merged["phi"])) and np.all(np.isfinite(merged["expected_BF"]))   # all 2048 prefixes were fitted
    fin = edlib.refset_finalize(merged)
    for name in st.dtype.names:
        x, y = fin["summary.stats"][name], st[name]
        assert np.array_equal(x.view("u4" if x.dtype.itemsize == 4 else "u8"), y.view("u4" if y.dtype.itemsize == 4 else "u8")), name
    assert fin["reference.choice"] == whole["reference.choice"]
    # n.bins.reduced = 10000, the vignette's setting: R's seq() thinning
    red = edlib.select_reference_set(test, refs, bin_length=length, n_bins_reduced=10000)
    assert red["n.bins"] == len(ro.r_seq_thin(exp["n_bins"], 10000))
    del refs
    torch.cuda.empty_cache()
