"""CPU tests of the checker against the committed golden vectors (no GPU, no /root/reference).

tests/golden/sf_ref.npz holds inputs and outputs of the REFERENCE'S OWN special-function sources
(src/beta.c, src/VP_gamma.c, src/VP_log.c, src/VP_psi.c, src/VP_zeta.c compiled as they lie into
oracle/_ref; generator: tests/golden/make_golden.py).  The libm flavour of the checker must reproduce
them bit for bit; the portable flavour (the device's bit-exact target) within the north-star tolerance.
"""
import hashlib
import itertools
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-10


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def test_libm_flavour_bitwise_equals_reference_build(oracle):
    g = np.load(os.path.join(G, "sf_ref.npz"))
    assert np.array_equal(bits(oracle.lnbeta(g["lnbeta_x"], g["lnbeta_y"], oracle.LIBM)), bits(g["lnbeta"]))
    assert np.array_equal(bits(oracle.sf("lngamma_e", g["lngamma_x"], oracle.LIBM)[0]), bits(g["lngamma"]))
    assert np.array_equal(bits(oracle.sf("gammastar", g["gammastar_x"], oracle.LIBM)[0]), bits(g["gammastar"]))
    assert np.array_equal(bits(oracle.sf("log_1plusx", g["log_1plusx_x"], oracle.LIBM)[0]), bits(g["log_1plusx"]))


def test_portable_flavour_within_tolerance_of_reference_build(oracle):
    g = np.load(os.path.join(G, "sf_ref.npz"))
    got = oracle.lnbeta(g["lnbeta_x"], g["lnbeta_y"], oracle.PORTABLE)
    ref = g["lnbeta"]
    big = np.abs(ref) > 1e-3
    assert np.max(np.abs(got[big] - ref[big]) / np.abs(ref[big])) < REL_TOL
    assert np.max(np.abs(got[~big] - ref[~big]), initial=0.0) < 1e-13
    for name, key in (("lngamma_e", "lngamma"), ("gammastar", "gammastar"), ("log_1plusx", "log_1plusx")):
        got = oracle.sf(name, g[key + "_x"], oracle.PORTABLE)[0]
        ref = g[key]
        big = np.abs(ref) > 1e-6
        assert np.max(np.abs(got[big] - ref[big]) / np.abs(ref[big])) < REL_TOL, name


def test_checker_digamma_trigamma_vs_reference_build(oracle):
    g = np.load(os.path.join(G, "sf_ref.npz"))
    psi, psi1 = oracle.psi(g["psi_x"])
    assert np.max(np.abs(psi - g["psi"]) / np.maximum(np.abs(g["psi"]), 1e-3)) < 1e-13
    assert np.max(np.abs(psi1 - g["psi_1"]) / np.abs(g["psi_1"])) < 1e-14


def test_hmm_known_answers_from_the_reference_docs(oracle):
    """reference R/tools.R:74-85: 'note the final 0 state, enforced by the code' and, with an identity
    transition matrix, 'we can check that no call is made'.  The call rows are the reference's own
    output as recorded in SURVEY.md section 8c (G3)."""
    T = np.full((3, 3), 1 / 3)
    ll = np.array([[0, -10, -10]] * 3 + [[-10, -10, 0]] * 3 + [[-10, 0, -10]] * 4, dtype=float)
    p, c = oracle.hmm(T, ll, np.arange(1, 11), 1)
    assert p.tolist() == [0, 0, 0, 2, 2, 2, 1, 1, 1, 0]
    assert c.tolist() == [[4, 6, 2, 3], [4, 9, 1, 3]]      # second call inherits the first one's start (quirk)
    p, c = oracle.hmm(np.eye(3), ll, np.arange(1, 11), 1)
    assert p.tolist() == [0] * 10 and len(c) == 0
    # forced ends: all-deletion emissions still start and finish in state 0
    ll = np.array([[-10, 0, -10]] * 5, dtype=float)
    p, c = oracle.hmm(T, ll, np.arange(1, 6), 1)
    assert p.tolist() == [0, 1, 1, 1, 0]
    assert oracle.hmm(np.full((2, 2), .5), ll[:, :2], np.arange(1, 6), 1, nstates=2) is None   # src/hmm.cpp:37-40


def _brute_force(T, ll, pos, L):
    """Best path by enumeration with the reference's scoring (src/hmm.cpp:62-79), first and last
    observation in state 0."""
    n = ll.shape[0]
    best, arg = -np.inf, None
    for mid in itertools.product(range(3), repeat=n - 2):
        st = (0,) + mid + (0,)
        sc = 0.0
        for i in range(1, n):
            d = np.exp(-(float(pos[i]) - float(pos[i - 1])) / L)
            j, k = st[i], st[i - 1]
            tr = T[0, j] if k == 0 else d * T[k, j] + (1 - d) * T[0, j]
            sc += ll[i, j] + np.log(tr)
        if sc > best:
            best, arg = sc, st
    return best, arg


def test_viterbi_path_is_optimal_by_enumeration(oracle):
    rng = np.random.default_rng(2)
    t = 1e-2
    T = np.array([[1 - t, t / 2, t / 2], [.5, .5, 0], [.5, 0, .5]])
    for _ in range(20):
        n = int(rng.integers(3, 9))
        ll = rng.normal(-2, 3, (n, 3))
        pos = np.cumsum(rng.integers(1, 5000, n)).astype(np.int32)
        path, _ = oracle.hmm(T, ll, pos, 3000.0)
        best, arg = _brute_force(T, ll, pos, 3000.0)
        sc = 0.0
        for i in range(1, n):
            d = np.exp(-(float(pos[i]) - float(pos[i - 1])) / 3000.0)
            j, k = path[i], path[i - 1]
            tr = T[0, j] if k == 0 else d * T[k, j] + (1 - d) * T[0, j]
            sc += ll[i, j] + np.log(tr)
        assert path[0] == 0 and path[-1] == 0
        assert abs(sc - best) < 1e-9, (path, arg)


def test_get_loglike_matrix_edge_rows(oracle):
    """Edge cases listed in SURVEY.md 8c (G2), with the behaviour the survey's probe of the reference recorded:
    total = 0 -> [0,0,0] exactly; expected = 0 -> [0,0,0] with GSL error events."""
    phi = np.array([0.005, 0.005, 0.005, 1e-9, 0.5, 0.002, 0.01, 0.005, 0.005])
    e = np.array([0.2, 0.2, 0.2, 0.1, 0.1, 0.12, 1e-6, 0.2, 0.0])
    tot = np.array([0, 500, 500, 900, 40, 2_000_000, 1000, 800, 10], dtype=np.int32)
    obs = np.array([0, 0, 500, 95, 3, 240_000, 0, 150, 3], dtype=np.int32)
    for fl in (oracle.LIBM, oracle.PORTABLE):
        L, nerr = oracle.get_loglike_matrix(phi, e, tot, obs, 1.0, fl)
        assert np.all(L[0] == 0.0)
        assert np.all(L[8] == 0.0) and nerr == 6        # 3 states x 2 lnbeta calls on NaN shape parameters
        assert np.all(np.isfinite(L[:8]))
        assert np.all(L[1:8] < 0)
    # the normal state is the most likely one when obs/total equals the expected proportion
    L, _ = oracle.get_loglike_matrix(0.005, 0.2, np.array([1000], np.int32), np.array([200], np.int32), 1.0, oracle.LIBM)
    assert L[0, 1] > L[0, 0] and L[0, 1] > L[0, 2]
    # mixture (prop.tumor) moves the CNV states toward the normal one
    L5, _ = oracle.get_loglike_matrix(0.005, 0.2, np.array([1000], np.int32), np.array([200], np.int32), 0.5, oracle.LIBM)
    assert L5[0, 0] > L[0, 0] and L5[0, 2] > L[0, 2] and L5[0, 1] == L[0, 1]


def _config1():
    d = np.load(os.path.join(G, "exomecount_chr1.npz"))
    return d["start"], d["end"], d["counts"]


def test_config1_bundled_data_matches_reference_probe(oracle):
    """BASELINE.json configs[0]: the reference's bundled data/ExomeCount.RData (fixture exomecount_chr1.npz).
    SURVEY.md 8c (G4) records what the reference's own compiled C gave for Exome1 vs Exome2+3+4:
    eta = -1.36727, phi = 0.0049568, 26 320 / 121 / 108 states over the 26 549 padded observations
    (= 26 318 / 121 / 108 over the exons), 25 calls; derived facts: sums, correlation, 3 791 empty exons."""
    start, end, counts = _config1()
    assert counts.shape == (26547, 4)
    assert counts[:, 0].sum() == 2_621_287 and counts[:, 1:].sum() == 10_427_365
    test = counts[:, 0].astype(np.int32)
    ref = counts[:, 1:].sum(axis=1).astype(np.int32)
    assert int(np.sum(test + ref == 0)) == 3791
    assert abs(np.corrcoef(test, ref)[0, 1] - 0.99274) < 5e-6
    phi, p, _, _ = oracle.fit_mle(test, ref)
    # the survey's (eta, phi) came from a scipy Nelder-Mead search: agreement to its tolerance, not to 1e-10
    assert abs(np.log(p / (1 - p)) - (-1.36727)) < 5e-6 and abs(phi - 0.0049568) < 5e-7
    L, nerr = oracle.get_loglike_matrix(phi, p, test + ref, test, 1.0, oracle.LIBM)
    path, calls = oracle.callcnvs(L, np.array([0, test.size], np.int32), start, end)
    assert nerr == 0
    assert np.bincount(path, minlength=3).tolist() == [26318, 121, 108]
    assert len(calls) == 25


def test_config1_regression_all_four_samples(oracle):
    start, end, counts = _config1()
    exp = np.load(os.path.join(G, "config1_expected.npz"))
    summ = json.load(open(os.path.join(G, "config1_summary.json")))
    chrom_off = np.array([0, counts.shape[0]], np.int32)
    for i in range(4):
        test = counts[:, i].astype(np.int32)
        ref = (counts.sum(axis=1) - counts[:, i]).astype(np.int32)
        phi, p = float(exp["phi%d" % i]), float(exp["p%d" % i])
        f_phi, f_p, _, _ = oracle.fit_mle(test, ref)
        assert abs(f_phi - phi) / phi < 1e-10 and abs(f_p - p) / p < 1e-10
        L, _ = oracle.get_loglike_matrix(phi, p, test + ref, test, 1.0, oracle.LIBM)
        assert hashlib.sha256(np.ascontiguousarray(L).tobytes()).hexdigest() == summ["sample%d" % (i + 1)]["loglik_sha256"]
        assert np.array_equal(bits(L[np.arange(0, test.size, 97)]), bits(exp["ll_rows%d" % i]))
        path, calls = oracle.callcnvs(L, chrom_off, start, end)
        assert np.array_equal(path, exp["path%d" % i]) and np.array_equal(calls, exp["calls%d" % i])
        # level C concordance: the portable flavour -- what the GPU evaluates -- gives
        # the same path and calls as the reference's arithmetic on the real data
        Lp, _ = oracle.get_loglike_matrix(phi, p, test + ref, test, 1.0, oracle.PORTABLE)
        pp, cp = oracle.callcnvs(Lp, chrom_off, start, end)
        assert np.array_equal(pp, path) and np.array_equal(cp, calls)
        nz = np.abs(L) > 0
        assert np.max(np.abs(Lp[nz] - L[nz]) / np.abs(L[nz])) < REL_TOL


def test_lngamma_sgn_negative_arguments_against_reference_vectors(oracle):
    """tests/golden/sf_ref_negative.npz: gsl_sf_lngamma_sgn_e of the reference build for x < 0 (reflection, lngamma_sgn_sing)."""
    g = np.load(os.path.join(G, "sf_ref_negative.npz"))
    v, s, st = oracle.lngamma_sgn(g["lngamma_sgn_x"], oracle.LIBM)
    assert np.all(st == 0) and np.array_equal(s, g["lngamma_sgn_sign"]) and np.array_equal(bits(v), bits(g["lngamma_sgn"]))
    pv, ps, _ = oracle.lngamma_sgn(g["lngamma_sgn_x"], oracle.PORTABLE)
    ref = g["lngamma_sgn"]
    big = np.abs(ref) > 1e-3
    assert np.array_equal(ps, s) and np.max(np.abs(pv[big] - ref[big]) / np.abs(ref[big])) < REL_TOL


def test_lnbeta_error_sites_and_values_outside_the_domain(oracle):
    """What gsl_sf_lnbeta returns and which gsl_error() calls it makes (src/beta.c:44, :56, :59, :163; src/VP_gamma.c:803,
    :1239, :1283) for arguments outside the positive quadrant."""
    nan = float("nan")
    x = np.array([0.0, 3.0, -2.0, nan, nan, -0.5, -0.5, -1.5, -2.0 + 1e-3, 5.0, -0.25])
    y = np.array([3.0, 0.0, 1.5, nan, 5.0, -0.3, 3.0, 4.0, 7.5, -5.5, 0.25])
    for fl in (oracle.LIBM, oracle.PORTABLE):
        v, c = oracle.lnbeta_sites(x, y, fl)
        assert c[0] == 1 << 9 and c[1] == 1 << 9 and np.isnan(v[0]) and np.isnan(v[1])         # beta.c:56
        assert c[2] == 1 << 10 and np.isnan(v[2])                                              # beta.c:59
        assert c[3] == (1 | 1 << 3 | 1 << 6) and v[3] == 0.0                                   # three EROUND exits, value 0
        assert c[4] == (1 | 1 << 6) and abs(v[4] - np.log(24.0)) < 1e-14                       # lnbeta(NaN, 5) = lgamma(5)
        assert c[5] == 1 << 11 and np.isnan(v[5])                                              # G(-.5) G(-.3) / G(-.8): negative
        assert c[6] == 1 << 11 and np.isnan(v[6])                                              # G(-.5) < 0, others > 0
        assert c[7] == 0 and np.isfinite(v[7])                                                 # G(-1.5) > 0
        assert c[8] == 0 and np.isfinite(v[8])                                                 # next to -2: lngamma_sgn_sing
        assert c[9] == 1 << 11 or c[9] == 0
        assert c[10] == (2 << 6) and np.isnan(v[10])                                           # x + y == 0: VP_gamma.c:1239
    import math
    v, _ = oracle.lnbeta_sites(np.array([-1.5]), np.array([4.0]), oracle.LIBM)
    assert abs(v[0] - (math.lgamma(-1.5) + math.lgamma(4.0) - math.lgamma(2.5))) < 1e-13
