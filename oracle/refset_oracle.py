"""CPU restatement of select.reference.set (reference R/optimize_reference_set.R:53-148) and
get.power.betabinom (reference R/tools.R:128-166).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the R code cannot run here (no R), it fits its models with aod::betabin and evaluates
VGAM::dbetabinom.ab -- both third-party, absent from the reference tree -- and none of its examples
asserts a value.  This module restates the R logic line by line (R's quantile type 7, stable
order(decreasing=TRUE), median, round-half-even, which.max) on top of the checker's own MLE
(edoracle.fit_mle) and its libm-flavour lnbeta; the GPU implementation is compared with it by tolerance.
"""
import numpy as np

from . import edoracle as eo


def r_quantile(x, p):
    """quantile(x, p), type 7"""
    v = np.sort(np.asarray(x, dtype=np.float64))
    h = (v.size - 1) * p
    lo = int(np.floor(h))
    if lo + 1 >= v.size:
        return v[-1]
    return v[lo] + (h - lo) * (v[lo + 1] - v[lo])


def lchoose(n, k):
    """R's lchoose for integer n >= k >= 0 through lbeta (nmath/choose.c: lfastchoose)"""
    n = np.asarray(n, dtype=np.float64); k = np.asarray(k, dtype=np.float64)
    return -np.log(n + 1.0) - eo.lnbeta(n - k + 1.0, k + 1.0, eo.LIBM)


def dbetabinom_ab_log(x, size, a, b):
    """VGAM::dbetabinom.ab(log = TRUE) as documented: lchoose + lbeta(a + x, b + size - x) - lbeta(a, b)"""
    x = np.asarray(x, dtype=np.float64)
    return (lchoose(np.full_like(x, size), x) + eo.lnbeta(a + x, b + size - x, eo.LIBM)
            - eo.lnbeta(np.array([a]), np.array([b]), eo.LIBM)[0])


def get_power_binom(size, my_p, my_alt_p):
    """reference R/tools.R:137-142 (theory = TRUE): sum over 0:size of dbinom(x; alt) * log10 of the binomial likelihood ratio.
    dbinom through scipy (R's nmath is not in the reference tree: parity unpinned, tolerance-level)."""
    from scipy.stats import binom
    x = np.arange(0, int(size) + 1)
    la = binom.logpmf(x, int(size), my_alt_p)
    l0 = binom.logpmf(x, int(size), my_p)
    return float(np.sum(np.exp(la) * (np.log10(np.e) * (la - l0))))


def get_power_betabinom(size, my_phi, my_p, my_alt_p):
    """reference R/tools.R:128-166 with theory = FALSE, limit = FALSE"""
    a = my_p * (1 - my_phi) / my_phi
    b = (1 - my_p) * (1 - my_phi) / my_phi
    aa = my_alt_p * (1 - my_phi) / my_phi
    ab = (1 - my_alt_p) * (1 - my_phi) / my_phi
    x = np.arange(0, int(size) + 1, dtype=np.float64)
    la = dbetabinom_ab_log(x, size, aa, ab)
    l0 = dbetabinom_ab_log(x, size, a, b)
    pr = np.exp(la)
    return float(np.sum(pr * (np.log10(np.e) * (la - l0))))


def r_seq_thin(length, n_reduced):
    """0-based positions of x[seq(1, length, length / n_reduced)]: R's seq.default for a fractional `by` computes
    n <- as.integer((to - from) / by + 1e-10) and from + (0:n) * by, clipped with pmin(., to); fractional subscripts
    truncate (R/optimize_reference_set.R:86)."""
    by = length / n_reduced
    n = int((length - 1.0) / by + 1e-10)
    x = np.minimum(1.0 + np.arange(n + 1, dtype=np.float64) * by, float(length))
    return x.astype(np.int64) - 1


def select_reference_set(test_counts, reference_counts, bin_length=None, n_bins_reduced=0):
    """Returns dict(order, correlations, expected_BF, phi, RatioSd, mean_p, median_depth, n_chosen, n_bins).
    Arrays are in the reference's row order (decreasing correlation); entries the R loop never reaches are NaN."""
    test = np.asarray(test_counts, dtype=np.float64)
    refs = np.asarray(reference_counts, dtype=np.float64)
    E, R = refs.shape
    if np.sum(test > 2) < 5:                                              # :57-61
        return {"n_chosen": 1, "order": np.arange(R)}
    L = np.ones(E) if bin_length is None else np.asarray(bin_length, dtype=np.float64)
    total = refs.sum(axis=1) + test                                       # :79
    q = r_quantile(total[total > 30], 0.9)                                # :80
    sel = np.where((total > 30) & (L >= r_quantile(L, 0.05)) & (L <= r_quantile(L, 0.95)) & (total < q))[0]   # :82-85
    if 0 < n_bins_reduced < sel.size:                                     # :86
        sel = sel[r_seq_thin(sel.size, n_bins_reduced)]
    test = test[sel]; refs = refs[sel]; L = L[sel]
    n = sel.size
    w = test / (L * test.sum() / 1e6)
    corr = np.array([np.corrcoef(refs[:, r] / (L * refs[:, r].sum() / 1e6), w)[0, 1] for r in range(R)])     # :100
    order = np.argsort(-corr, kind="stable")                              # :101
    out = {k: np.full(R, np.nan) for k in ("expected_BF", "phi", "RatioSd", "mean_p", "median_depth")}
    out["order"] = order
    out["correlations"] = corr[order]
    out["n_bins"] = int(n)
    reference = np.zeros(n)
    for i in range(R):                                                    # :114
        reference = reference + refs[:, order[i]]
        phi, p, _, _ = eo.fit_mle(test.astype(np.int32), reference.astype(np.int32))   # :117-123 (aod::betabin stand-in)
        out["phi"][i] = phi
        out["mean_p"][i] = p
        out["median_depth"][i] = np.median(reference)
        out["RatioSd"][i] = np.mean(np.sqrt(1 + (test + reference - 1) * phi))
        if i + 1 > 2 and p < 0.05:                                        # :130
            break
        alt_odds = p / (1 - p) * 0.5
        alt_p = alt_odds / (1 + alt_odds)
        out["expected_BF"][i] = get_power_betabinom(np.round(out["median_depth"][i]), phi, p, alt_p)   # :135-139
    out["n_chosen"] = int(np.nanargmax(out["expected_BF"])) + 1           # :143 which.max
    return out


def select_reference_set_lean(test_counts, reference_counts, bin_length=None, n_bins_reduced=0, raw_prefixes=()):
    """The same restatement for BASELINE configs[4] sizes (500 000 bins x 2048 references): the count matrix stays int32
    and is walked in column blocks (the plain version above converts it to float64 whole: 8 GB + copies).  Runs the R
    loop with its early exit (rows the loop never reaches stay NaN) and, in addition, evaluates the RAW statistics of
    the cumulative references listed in `raw_prefixes` (0-based row indices; no early exit) -- what
    ed_select_reference_set_part returns for those rows.  Returns the dict of select_reference_set() plus
    'raw': {i: dict(phi, mean_p, median_depth, RatioSd, expected_BF)}."""
    test = np.asarray(test_counts, dtype=np.float64)
    refs = np.asarray(reference_counts)
    assert refs.dtype == np.int32 and refs.ndim == 2
    E, R = refs.shape
    L = np.ones(E) if bin_length is None else np.asarray(bin_length, dtype=np.float64)
    total = refs.sum(axis=1, dtype=np.int64).astype(np.float64) + test
    q = r_quantile(total[total > 30], 0.9)
    sel = np.where((total > 30) & (L >= r_quantile(L, 0.05)) & (L <= r_quantile(L, 0.95)) & (total < q))[0]
    if 0 < n_bins_reduced < sel.size:
        sel = sel[r_seq_thin(sel.size, n_bins_reduced)]
    test = test[sel]; L = L[sel]
    refs = refs[sel]                                                      # int32 (n, R)
    n = sel.size
    w = test / (L * test.sum() / 1e6)
    corr = np.empty(R)
    for c0 in range(0, R, 64):
        blk = refs[:, c0:c0 + 64].astype(np.float64)
        for j in range(blk.shape[1]):
            x = blk[:, j]
            corr[c0 + j] = np.corrcoef(x / (L * x.sum() / 1e6), w)[0, 1]
    order = np.argsort(-corr, kind="stable")
    out = {k: np.full(R, np.nan) for k in ("expected_BF", "phi", "RatioSd", "mean_p", "median_depth")}
    out.update(order=order, correlations=corr[order], n_bins=int(n), raw={})
    want_raw = set(int(i) for i in raw_prefixes)
    last_raw = max(want_raw) if want_raw else -1
    ti = test.astype(np.int32)

    def stats(reference):
        phi, p, _, _ = eo.fit_mle_hist(ti, reference.astype(np.int32))    # the long-double MLE on sufficient statistics
        med = float(np.median(reference))
        rsd = float(np.mean(np.sqrt(1 + (test + reference - 1) * phi)))
        alt_odds = p / (1 - p) * 0.5
        bf = get_power_betabinom(np.round(med), phi, p, alt_odds / (1 + alt_odds))
        return phi, p, med, rsd, bf

    reference = np.zeros(n)
    in_loop = True
    for i in range(R):
        if not in_loop and i > last_raw:
            break
        reference = reference + refs[:, order[i]]
        if not in_loop and i not in want_raw:
            continue
        phi, p, med, rsd, bf = stats(reference)
        if i in want_raw:
            out["raw"][i] = dict(phi=phi, mean_p=p, median_depth=med, RatioSd=rsd, expected_BF=bf)
        if in_loop:
            out["phi"][i], out["mean_p"][i], out["median_depth"][i], out["RatioSd"][i] = phi, p, med, rsd
            if i + 1 > 2 and p < 0.05:
                in_loop = False                                           # :130 break: expected.BF[i] stays NA
                continue
            out["expected_BF"][i] = bf
    out["n_chosen"] = int(np.nanargmax(out["expected_BF"])) + 1
    return out
