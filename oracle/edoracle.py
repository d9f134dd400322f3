"""ctypes front-end of the CPU checker (oracle/libed_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (exomedepth_amd) never does.  See oracle/ed_oracle.c for what is restated and how
it is pinned to the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
LIBM, PORTABLE = 0, 1

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_bp = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")


def build(force=False):
    """(Re)build the checker -- and oracle/_ref when /root/reference is present -- with make."""
    so = os.path.join(_HERE, "libed_oracle.so")
    if force or not os.path.exists(so) or os.path.exists("/root/reference/src/beta.c"):
        subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libed_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        for name in ("edo_plog_v", "edo_pexp_v", "edo_psin_v", "edo_psin_any_v"):
            getattr(L, name).argtypes = [C.c_long, _dp, _dp]
            getattr(L, name).restype = None
        L.edo_dtab.argtypes = [C.c_double, C.c_long, _dp]
        L.edo_ddlog_v.argtypes = [C.c_long, _dp, _dp, _dp]
        L.edo_dtab_combine_v.argtypes = [C.c_long, _dp, _dp, _dp, _dp]
        L.edo_dtab_lg0.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.edo_dtab_lg0.restype = None
        L.edo_dtab_tail_v.argtypes = [C.c_double, C.c_long, _dp, _dp]
        L.edo_dtab_tail_v.restype = None
        L.edo_lnbeta_v.argtypes = [C.c_int, C.c_long, _dp, _dp, _dp]
        L.edo_lnbeta_v.restype = C.c_long
        L.edo_sf_v.argtypes = [C.c_int, C.c_int, C.c_long, _dp, _dp, _ip]
        L.edo_sf_v.restype = None
        L.edo_get_loglike_matrix.argtypes = [C.c_int, _dp, _dp, _ip, _ip, C.c_long, C.c_double, _dp]
        L.edo_get_loglike_matrix.restype = C.c_long
        L.edo_hmm.argtypes = [C.c_int, C.c_long, _dp, _dp, _ip, C.c_double, _dp, _dp, C.c_long]
        L.edo_hmm.restype = C.c_long
        L.edo_callcnvs.argtypes = [_dp, C.c_long, _ip, C.c_int, _ip, _ip, C.c_double, C.c_double, _bp, _dp, C.c_long]
        L.edo_callcnvs.restype = C.c_long
        L.edo_ref_open.argtypes = [C.c_char_p]
        L.edo_ref_open.restype = C.c_int
        L.edo_ref_call1.argtypes = [C.c_char_p, C.c_long, _dp, _dp]
        L.edo_ref_call1.restype = C.c_int
        L.edo_ref_call2.argtypes = [C.c_char_p, C.c_long, _dp, _dp, _dp]
        L.edo_ref_call2.restype = C.c_int
        L.edo_now.restype = C.c_double
        L.edo_ref_lngamma_sgn.argtypes = [C.c_long, _dp, _dp, _dp, _ip]
        L.edo_ref_lngamma_sgn.restype = C.c_int
        L.edo_lngamma_sgn_v.argtypes = [C.c_int, C.c_long, _dp, _dp, _dp, _ip]
        L.edo_lngamma_sgn_v.restype = None
        L.edo_lnbeta_sites_v.argtypes = [C.c_int, C.c_long, _dp, _dp, _dp, _ip]
        L.edo_lnbeta_sites_v.restype = None
        L.edo_psi_v.argtypes = [C.c_long, _dp, _dp, _dp]
        L.edo_psi_v.restype = None
        L.edo_fit_mle.argtypes = [_ip, _ip, C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.edo_fit_mle.restype = C.c_int
        L.edo_fit_mle_hist.argtypes = [_ip, _ip, C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.edo_fit_mle_hist.restype = C.c_int
        L.edo_fit_mle_groups.argtypes = [_ip, _ip, _ip, C.c_long, C.c_int, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.edo_fit_mle_groups.restype = C.c_int
        L.edo_fit_mle_cov.argtypes = [_ip, _ip, _dp, C.c_long, C.c_int, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.edo_fit_mle_cov.restype = C.c_int
        L.edo_fit_nm.argtypes = [_ip, _ip, C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.edo_fit_nm.restype = C.c_int
        _LIB = L
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def plog(x):
    x = _f64(x); out = np.empty_like(x); lib().edo_plog_v(x.size, x, out); return out


def pexp(x):
    x = _f64(x); out = np.empty_like(x); lib().edo_pexp_v(x.size, x, out); return out


def psin(x):
    x = _f64(x); out = np.empty_like(x); lib().edo_psin_v(x.size, x, out); return out


def psin_any(x):
    x = _f64(x); out = np.empty_like(x); lib().edo_psin_any_v(x.size, x, out); return out


def dtab(x0, n):
    """D(x0, k) = lgamma(x0 + k) - lgamma(x0), k = 0..n-1, as the table-driven emission mode tabulates it (csrc/ed_dtab.h)"""
    out = np.empty(int(n)); lib().edo_dtab(float(x0), int(n), out); return out


def dtab_lg0(x0):
    """lgamma(x0) as the double-double (hi, lo) the Stirling tails subtract (csrc/ed_dtab.h: ed_dtab_lg0)"""
    hi, lo = C.c_double(0), C.c_double(0)
    lib().edo_dtab_lg0(float(x0), C.byref(hi), C.byref(lo))
    return hi.value, lo.value


def dtab_tail(x0, k):
    """D(x0, k) = lgamma(x0 + k) - lgamma(x0) for counts k >= 64 from Stirling's series (csrc/ed_dtab.h: ed_dtab_tail)"""
    k = _f64(k); out = np.empty_like(k); lib().edo_dtab_tail_v(float(x0), k.size, k, out); return out


def ddlog(x):
    x = _f64(x); hi = np.empty_like(x); lo = np.empty_like(x); lib().edo_ddlog_v(x.size, x, hi, lo); return hi, lo


def dtab_combine(d1, d2, d3):
    d1 = _f64(d1); d2 = _f64(d2); d3 = _f64(d3); out = np.empty_like(d1); lib().edo_dtab_combine_v(d1.size, d1, d2, d3, out); return out


def lnbeta(x, y, flavour=PORTABLE):
    x = _f64(x); y = _f64(y); out = np.empty_like(x)
    lib().edo_lnbeta_v(flavour, x.size, x, y, out)
    return out


_SF = {"lngamma": 0, "gammastar": 1, "log_1plusx": 2, "lngamma_e": 3}


def sf(which, x, flavour=PORTABLE):
    x = _f64(x); out = np.empty_like(x); st = np.zeros(x.size, dtype=np.int32)
    lib().edo_sf_v(flavour, _SF[which], x.size, x, out, st)
    return out, st


def lngamma_sgn(x, flavour=PORTABLE):
    """gsl_sf_lngamma_sgn_e of the checker: (value, sign, status)"""
    x = _f64(x); v = np.empty_like(x); sg = np.empty_like(x); st = np.zeros(x.size, dtype=np.int32)
    lib().edo_lngamma_sgn_v(flavour, x.size, x, v, sg, st)
    return v, sg, st


def lnbeta_sites(x, y, flavour=PORTABLE):
    """(value, error sites) of gsl_sf_lnbeta_e; sites coded as exomedepth_amd/csrc/ed_sf_dev.hpp::lnbeta_sites"""
    x = _f64(x); y = _f64(y); v = np.empty_like(x); c = np.zeros(x.size, dtype=np.int32)
    lib().edo_lnbeta_sites_v(flavour, x.size, x, y, v, c)
    return v, c


def ref_lngamma_sgn(x):
    """gsl_sf_lngamma_sgn_e of the reference build (oracle/_ref): (value, sign, status); error-free arguments only"""
    _ref_open()
    x = _f64(x); v = np.empty_like(x); sg = np.empty_like(x); st = np.zeros(x.size, dtype=np.int32)
    if lib().edo_ref_lngamma_sgn(x.size, x, v, sg, st) != 0:
        raise RuntimeError("gsl_sf_lngamma_sgn_e not found in the reference build")
    return v, sg, st


def get_loglike_matrix(phi, expected, total, observed, mixture=1.0, flavour=PORTABLE):
    """reference src/CNV_estimate.cpp:52-85 -> (n,3) array, columns (deletion, normal, duplication)."""
    total = _i32(total); n = total.size
    phi = _f64(np.broadcast_to(phi, (n,))); expected = _f64(np.broadcast_to(expected, (n,)))
    observed = _i32(observed)
    out = np.empty((3, n), dtype=np.float64)  # column-major n x 3
    nerr = lib().edo_get_loglike_matrix(flavour, phi, expected, total, observed, n, float(mixture), out)
    return out.T, nerr


def hmm(transitions, loglikelihood, positions, expected_cnv_length, nstates=3):
    """reference R/tools.R:88-103 + src/hmm.cpp:18-167.  loglikelihood: (nobs,3) in HMM order
    (normal, deletion, duplication).  Returns (path int array, calls (ncalls,4) float array)."""
    T = np.asarray(transitions, dtype=np.float64)
    if T.shape[0] != T.shape[1]:
        raise ValueError("Transition matrix is not square")
    ll = np.asarray(loglikelihood, dtype=np.float64)
    positions = _i32(positions)
    if positions.size != ll.shape[0]:
        raise ValueError("The number of positions are not matching the number of rows of the likelihood matrix")
    nobs = ll.shape[0]
    Tc = _f64(T.T.ravel())       # column-major
    llc = _f64(ll.T.ravel())     # column-major nobs x 3
    path = np.empty(nobs, dtype=np.float64)
    cap = max(nobs, 1)
    calls = np.zeros((cap, 4), dtype=np.float64)
    nc = lib().edo_hmm(int(nstates), nobs, Tc, llc, positions, float(expected_cnv_length), path, calls, cap)
    if nc < 0:
        return None
    return path.astype(np.int64), calls[:nc].copy()


MARGIN_THRESHOLDS = (1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-3)


def callcnvs_margins(likelihood, chrom_off, start, end, transition_probability=1e-4, expected_cnv_length=50000.0, acc=None):
    """Decision margins along the Viterbi path of one sample (edo_callcnvs_margins): accumulates into / returns a dict with `decisions` (on-path
    decisions between two finite candidates), `ties` (margin exactly 0), `below` {threshold: count of non-zero margins below it}, `min_margin`,
    `scale_at_min` (|best candidate| there)."""
    ll = np.asarray(likelihood, dtype=np.float64)
    n = ll.shape[0]
    llc = _f64(ll.T.ravel())
    chrom_off = _i32(chrom_off)
    thr = np.array(MARGIN_THRESHOLDS, dtype=np.float64)
    below = np.zeros(thr.size, dtype=np.int64)
    ties, dec = C.c_long(0), C.c_long(0)
    mn, sc = C.c_double(np.inf), C.c_double(0.0)
    L = lib()
    if not getattr(L, "_margins_typed", False):
        L.edo_callcnvs_margins.argtypes = [_dp, C.c_long, _ip, C.c_int, _ip, _ip, C.c_double, C.c_double, _dp, C.c_int,
                                           np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS"), C.POINTER(C.c_long), C.POINTER(C.c_long),
                                           C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.edo_callcnvs_margins.restype = None
        L._margins_typed = True
    L.edo_callcnvs_margins(llc, n, chrom_off, chrom_off.size - 1, _i32(start), _i32(end), float(transition_probability), float(expected_cnv_length),
                           thr, thr.size, below, C.byref(ties), C.byref(dec), C.byref(mn), C.byref(sc))
    out = acc if acc is not None else {"decisions": 0, "ties": 0, "below": {("%g" % t): 0 for t in MARGIN_THRESHOLDS}, "min_margin": float("inf"), "scale_at_min": 0.0}
    out["decisions"] += dec.value
    out["ties"] += ties.value
    for t, b in zip(MARGIN_THRESHOLDS, below):
        out["below"]["%g" % t] += int(b)
    if mn.value < out["min_margin"]:
        out["min_margin"], out["scale_at_min"] = mn.value, sc.value
    return out


def callcnvs(likelihood, chrom_off, start, end, transition_probability=1e-4, expected_cnv_length=50000.0):
    """reference R/class_definition.R:343-374, :408-414 on pre-ordered exons.  likelihood (n,3) in
    (deletion, normal, duplication) order.  Returns (path int8[n], calls (ncalls,4))."""
    ll = np.asarray(likelihood, dtype=np.float64)
    n = ll.shape[0]
    llc = _f64(ll.T.ravel())
    chrom_off = _i32(chrom_off)
    path = np.zeros(n, dtype=np.int8)
    cap = n + 8
    calls = np.zeros((cap, 4), dtype=np.float64)
    nc = lib().edo_callcnvs(llc, n, chrom_off, chrom_off.size - 1, _i32(start), _i32(end),
                            float(transition_probability), float(expected_cnv_length), path, calls, cap)
    return path, calls[:nc].copy()


def psi(x):
    """(digamma, trigamma) of the checker (long double inside)."""
    x = _f64(x); a = np.empty_like(x); b = np.empty_like(x)
    lib().edo_psi_v(x.size, x, a, b)
    return a, b


def fit_mle(test, ref):
    """High-precision MLE of (phi, p) for  cbind(test, reference) ~ 1  (parity unpinned: see edo_fit.inc)."""
    test = _i32(test); ref = _i32(ref)
    phi, p, ll = C.c_double(), C.c_double(), C.c_double()
    it = lib().edo_fit_mle(test, ref, test.size, C.byref(phi), C.byref(p), C.byref(ll))
    return phi.value, p.value, ll.value, it


def fit_mle_hist(test, ref):
    """fit_mle on sufficient statistics (sums over distinct count values): same estimator, ~100x faster on long columns."""
    test = _i32(test); ref = _i32(ref)
    phi, p, ll = C.c_double(), C.c_double(), C.c_double()
    it = lib().edo_fit_mle_hist(test, ref, test.size, C.byref(phi), C.byref(p), C.byref(ll))
    return phi.value, p.value, ll.value, it


def fit_mle_groups(test, ref, grp, n_groups):
    """High-precision MLE of (phi_1..phi_B, p) for  cbind(test, reference) ~ 1, random = ~ depth.quant
    (phi.bins > 1, reference R/class_definition.R:135-139; parity unpinned).  grp: 0-based group of each row."""
    test = _i32(test); ref = _i32(ref); grp = _i32(grp)
    phi = np.zeros(n_groups)
    p, ll = C.c_double(), C.c_double()
    it = lib().edo_fit_mle_groups(test, ref, grp, test.size, n_groups, phi, C.byref(p), C.byref(ll))
    if it < 0:
        raise ValueError("edo_fit_mle_groups: nothing to fit (%d)" % it)
    return phi, p.value, ll.value, it


def fit_mle_cov(test, ref, X):
    """High-precision MLE of (beta_0..beta_K, phi) for  cbind(test, reference) ~ x1 + ... + xK  (a `data` frame with
    covariates, reference R/class_definition.R:118; parity unpinned).  X: (n, K) covariates."""
    test = _i32(test); ref = _i32(ref)
    X = _f64(np.asarray(X, dtype=np.float64).reshape(test.size, -1))
    K = X.shape[1]
    beta = np.zeros(K + 1)
    phi, ll = C.c_double(), C.c_double()
    it = lib().edo_fit_mle_cov(test, ref, X, test.size, K, beta, C.byref(phi), C.byref(ll))
    if it < 0:
        raise ValueError("edo_fit_mle_cov: nothing to fit (%d)" % it)
    return beta, phi.value, ll.value, it


def fit_nm(test, ref, with_status=False):
    """aod::betabin's documented procedure (its objective, R's nmmin from the glm start, optim()'s defaults): (phi, p, function
    evaluations[, nmmin's fail code]).  A stand-in -- aod is not in the reference tree."""
    test = _i32(test); ref = _i32(ref)
    phi, p, ne = C.c_double(), C.c_double(), C.c_int()
    fail = lib().edo_fit_nm(test, ref, test.size, C.byref(phi), C.byref(p), C.byref(ne))
    return (phi.value, p.value, ne.value, fail) if with_status else (phi.value, p.value, ne.value)


# ---- the reference's own special functions, compiled as they lie (container only) ----
def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libgslsf_ref.so"))


def _ref_open():
    if lib().edo_ref_open(os.path.join(_HERE, "_ref", "libgslsf_ref.so").encode()) != 0:
        raise RuntimeError("oracle/_ref/libgslsf_ref.so not loadable")


def ref_call1(name, x):
    _ref_open()
    x = _f64(x); out = np.empty_like(x)
    if lib().edo_ref_call1(name.encode(), x.size, x, out) != 0:
        raise RuntimeError("reference symbol %s not found" % name)
    return out


def ref_call2(name, x, y):
    _ref_open()
    x = _f64(x); y = _f64(y); out = np.empty_like(x)
    if lib().edo_ref_call2(name.encode(), x.size, x, y, out) != 0:
        raise RuntimeError("reference symbol %s not found" % name)
    return out
