"""TEST INFRASTRUCTURE (checker): numpy restatement of the depth-binned dispersion of the reference,
`phi.bins > 1` in the `ExomeDepth` initialiser (reference R/class_definition.R:120-147).

PARITY UNPINNED: the per-bin dispersions come from aod::betabin(random = ~ depth.quant), a third-party package
that is not in the reference tree; the checker uses the long-double MLE of the documented likelihood
(oracle/edo_fit.inc::edo_fit_mle_groups).  The binning, the interpolation and R's quantile()/seq()/approxfun()
semantics are restated from the R sources line by line.
"""
import numpy as np

from . import edoracle as eo


def r_quantile7(x, prob):
    """stats::quantile(x, prob, type = 7) for one probability (R's default type)."""
    x = np.sort(np.asarray(x, dtype=np.float64))
    n = x.size
    index = 1 + max(n - 1, 0) * prob
    lo = int(np.floor(index))
    hi = int(np.ceil(index))
    qs = x[lo - 1]
    if index > lo and x[hi - 1] != qs:
        h = index - lo
        qs = (1 - h) * qs + h * x[hi - 1]
    return qs


def r_seq_by(frm, to, by):
    """seq.default(from, to, by) for doubles with by > 0 (base R)."""
    delta = to - frm
    if delta == 0 and to == 0:
        return np.array([to])
    n = delta / by
    if not np.isfinite(n):
        raise ValueError("invalid '(to - from)/by' in seq(.)")
    if n < 0:
        raise ValueError("wrong sign in 'by' argument")
    dd = abs(delta) / max(abs(to), abs(frm))
    if dd < 100 * np.finfo(float).eps:
        return np.array([frm])
    n = int(n + 1e-10)
    x = frm + np.arange(n + 1) * by
    return np.minimum(x, to) if by > 0 else np.maximum(x, to)


def depth_bins(reference, phi_bins):
    """R/class_definition.R:124-133: complete.bins (phi.bins + 1 edges) and depth.quant (1-based level per exon)."""
    reference = np.asarray(reference, dtype=np.float64)
    q85 = r_quantile7(reference, 0.85)
    qmax = r_quantile7(reference, 1.0)
    if q85 == 0:
        bottom = np.array([0.0])                      # seq(0, 0, by = 0/(B-1)) = 0
    else:
        bottom = r_seq_by(0.0, q85, q85 / (phi_bins - 1))
    complete = np.concatenate([bottom, [qmax + 1]])
    quant = (reference[:, None] >= complete[None, :]).sum(axis=1)
    if np.unique(quant).size != phi_bins:
        raise ValueError("Binning did not happen properly")
    return complete, quant


def approx_linear(v, x, y):
    """approxfun(x, y, yleft = y[1], yright = y[n]) evaluated at v (R's C approx1, linear)."""
    v = np.asarray(v, dtype=np.float64)
    out = np.empty_like(v)
    n = x.size
    for k, vk in enumerate(v):
        if vk < x[0]:
            out[k] = y[0]
            continue
        if vk > x[n - 1]:
            out[k] = y[n - 1]
            continue
        i, j = 0, n - 1
        while i < j - 1:
            ij = (i + j) // 2
            if vk < x[ij]:
                j = ij
            else:
                i = ij
        if vk == x[j]:
            out[k] = y[j]
        elif vk == x[i]:
            out[k] = y[i]
        else:
            out[k] = y[i] + (y[j] - y[i]) * ((vk - x[i]) / (x[j] - x[i]))
    return out


def fit_bins(test, reference, phi_bins):
    """Returns (phi.estimates[B], expected, phi.linear[E], complete.bins[B+1])."""
    test = np.asarray(test)
    reference = np.asarray(reference)
    complete, quant = depth_bins(reference, phi_bins)
    phi_est, p, _, _ = eo.fit_mle_groups(test.astype(np.int32), reference.astype(np.int32), (quant - 1).astype(np.int32),
                                         phi_bins)
    mid = (complete[:phi_bins] + complete[1:phi_bins + 1]) / 2
    return phi_est, p, approx_linear(reference.astype(np.float64), mid, phi_est), complete
