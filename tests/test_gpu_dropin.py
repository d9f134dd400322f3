"""The two reference-shaped entries (ed_get_loglike_matrix, ed_hmm: csrc/eddropin.inc) at the call pattern of the unchanged S4 surface:
one .Call get_loglike_matrix per sample (reference R/class_definition.R:184-189) and one .Call C_hmm per chromosome and sample
(R/tools.R:97 <- R/class_definition.R:354-374) -- against the CPU checker, bit for bit.  Round 6: scratch kept between calls, packed
rows, a whole wave on the forward pass, parallel trace-back and call summary, log-transitions remembered per chain."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def _calls_array(res):
    c = res["calls"]
    return np.stack([c[k] for k in ("start.p", "end.p", "type", "nexons")], axis=1) if len(c) else np.zeros((0, 4))


def _random_T(rng, kind):
    if kind == "callcnvs":                       # R/class_definition.R:343-347
        t = 10.0 ** rng.uniform(-6, -1)
        return np.array([[1 - t, t / 2, t / 2], [.5, .5, 0], [.5, 0, .5]])
    if kind == "uniform":                        # R/tools.R:74
        return np.full((3, 3), 1 / 3)
    T = rng.random((3, 3)) + 0.01                # any row-stochastic matrix, zeros allowed
    T[rng.random((3, 3)) < 0.15] = 0.0
    T[:, 0] += 0.05
    return T / T.sum(axis=1, keepdims=True)


def _emissions(rng, nobs, kind):
    if kind == "runs":                           # long runs that favour one state: many calls, direct 1 -> 2 switches included
        ll = np.full((nobs, 3), -8.0)
        i = 0
        while i < nobs:
            n = int(rng.integers(1, 40)); s = int(rng.integers(0, 3))
            ll[i:i + n, s] = rng.uniform(-1.0, 0.0)
            i += n
        return ll
    ll = rng.normal(-3, 2, (nobs, 3))
    if kind == "ties":                           # values from a small set: equal candidates, the first strict maximum decides
        ll = rng.integers(-4, 1, (nobs, 3)).astype(np.float64)
    ll[rng.random(nobs) < 0.02, 1] = -np.inf
    ll[rng.random(nobs) < 0.01, 2] = -np.inf
    ll[rng.random(nobs) < 0.01] = 0.0
    ll[rng.random(nobs) < 0.002] = -np.inf     # all three states impossible (the reference reads from_where = -1 there: the build's defined 0)
    return ll


@pytest.mark.parametrize("nobs", [1, 2, 3, 16, 17, 18, 32, 33, 34, 63, 64, 65, 66, 97, 1000, 8302, 19354, 32769, 32786, 70001])
def test_hmm_chain_lengths(edlib, oracle, nobs):
    """every tile / word / chunk boundary of the forward pass (32-step tiles, 16-step words) and of the trace-back (2 048-word chunks)"""
    rng = np.random.default_rng(1000 + nobs)
    for kind_T, kind_e in (("callcnvs", "normal"), ("any", "runs"), ("uniform", "ties")):
        T = _random_T(rng, kind_T)
        ll = _emissions(rng, nobs, kind_e)
        pos = np.cumsum(rng.integers(1, 30000, nobs)).astype(np.int32)
        L = float(10.0 ** rng.uniform(2, 6))
        got = edlib.viterbi_hmm(T, ll, pos, L)
        p, c = oracle.hmm(T, ll, pos, L)
        assert np.array_equal(got["Viterbi.path"], p), (nobs, kind_T, kind_e)
        assert np.array_equal(_calls_array(got), c), (nobs, kind_T, kind_e)


def test_hmm_remembers_a_chain_only_for_identical_inputs(edlib, oracle):
    """The log-transitions of a chain are kept between calls (same positions, matrix, length: CallCNVs' loop over samples).  Every change of an
    input must miss; a hit must give what a fresh evaluation gives.  More chains than the cache holds, visited twice."""
    from exomedepth_amd._lib import lib
    rng = np.random.default_rng(77)
    lib().ed_dropin_release()
    chains = []
    for k in range(80):                                          # (the cache keeps 64)
        nobs = int(rng.integers(40, 1500))
        chains.append((_random_T(rng, "callcnvs" if k % 2 else "any"), np.cumsum(rng.integers(1, 20000, nobs)).astype(np.int32), float(rng.choice([5e4, 1e3, 2e5]))))
    for rnd in range(2):
        for T, pos, L in chains:
            ll = _emissions(rng, pos.size, "runs" if rnd else "normal")
            got = edlib.viterbi_hmm(T, ll, pos, L)
            p, c = oracle.hmm(T, ll, pos, L)
            assert np.array_equal(got["Viterbi.path"], p) and np.array_equal(_calls_array(got), c)
    # one chain, one input changed at a time
    T, pos, L = chains[-1]
    ll = _emissions(rng, pos.size, "runs")
    variants = [(T, pos, L), (T, pos, L * 1.5), (T.T.copy() / T.T.sum(axis=1, keepdims=True), pos, L)]
    q = pos.copy(); q[pos.size // 2:] += 7
    variants.append((T, q, L))
    q2 = pos.copy(); q2[-1] += 1
    variants.append((T, q2, L))
    variants.append((T, pos[:-1], L))
    for Tv, pv, Lv in variants + variants:
        got = edlib.viterbi_hmm(Tv, ll[:pv.size], pv, Lv)
        p, c = oracle.hmm(Tv, ll[:pv.size], pv, Lv)
        assert np.array_equal(got["Viterbi.path"], p) and np.array_equal(_calls_array(got), c)
    lib().ed_dropin_release()
    got = edlib.viterbi_hmm(T, ll, pos, L)                       # after a release: a fresh scratch
    p, c = oracle.hmm(T, ll, pos, L)
    assert np.array_equal(got["Viterbi.path"], p) and np.array_equal(_calls_array(got), c)


def test_hmm_call_capacity(edlib, oracle):
    """calls_cap below the number of calls: n_calls reports them all, the first calls_cap rows are written, nothing beyond (src/hmm.cpp has no cap:
    the R shim hands nobs rows)"""
    from exomedepth_amd._lib import check, lib
    rng = np.random.default_rng(5)
    nobs = 6000
    T = _random_T(rng, "uniform")
    ll = np.full((nobs, 3), -30.0)                                # short, sharp runs: a call every few observations
    i = 0
    while i < nobs:
        n = int(rng.integers(2, 9))
        ll[i:i + n, int(rng.integers(0, 3))] = 0.0
        i += n
    pos = np.cumsum(rng.integers(1, 3000, nobs)).astype(np.int32)
    p, c = oracle.hmm(T, ll, pos, 1000.0)
    assert len(c) > 300                                           # (more than the first copy's 256 rows: the second copy is exercised below)
    Tc = np.ascontiguousarray(T.T.ravel()); llc = np.ascontiguousarray(ll.T.ravel())
    for cap in (0, 1, 7, 256, 257, len(c) - 1, len(c), len(c) + 5, nobs):
        path = np.empty(nobs); calls = np.full((4, max(cap, 1) + 3), -7.0); nc = C.c_int64(-1)
        cc = np.ascontiguousarray(calls[:, :max(cap, 1)])
        check(lib().ed_hmm(3, nobs, C.c_void_p(Tc.ctypes.data), C.c_void_p(llc.ctypes.data), C.c_void_p(pos.ctypes.data), 1000.0,
                           C.c_void_p(path.ctypes.data), C.c_void_p(cc.ctypes.data) if cap else None, cap, C.byref(nc)))
        assert nc.value == len(c)
        assert np.array_equal(path.astype(np.int64), p)
        k = min(cap, len(c))
        assert np.array_equal(cc[:, :k].T, c[:k])
        if cap > k:
            assert np.all(cc[:, k:cap] == -7.0)


def _rows(rng, n):
    phi = 10.0 ** rng.uniform(-4, -1, n)
    e = rng.uniform(0.02, 0.6, n)
    tot = rng.integers(0, 3000, n).astype(np.int32)
    obs = (tot * np.clip(rng.normal(e, 0.05), 0, 1)).astype(np.int32)
    return phi, e, tot, obs


@pytest.mark.parametrize("n", [1, 2, 255, 257, 65535, 65536, 65537, 65541, 200_000, 262_147, 1_000_003])
def test_get_loglike_matrix_sizes(edlib, oracle, n):
    """one piece below 65 536 rows, four pipelined pieces from there on (odd sizes: the pieces' int32 parts must stay aligned); the matrix
    comes back column-major n x 3 with the error count behind the last piece"""
    rng = np.random.default_rng(n)
    phi, e, tot, obs = _rows(rng, n)
    if n > 10:
        e[n // 3] = 0.0                                            # a row with GSL error events (src/beta.c:54-60)
    for mix in (1.0, 0.7):
        got, nerr = edlib.get_loglike_matrix(phi, e, tot, obs, mixture=mix, return_errors=True)
        exp, oerr = oracle.get_loglike_matrix(phi, e, tot, obs, mixture=mix, flavour=oracle.PORTABLE)
        same = (bits(got) == bits(exp)) | (np.isnan(got) & np.isnan(exp))
        assert same.all()
        assert nerr == oerr


def test_one_sample_through_the_unchanged_surface(edlib, oracle):
    """new('ExomeDepth') + CallCNVs() of ONE sample as the reference makes them: 1 call of get_loglike_matrix on all exons (phi, expected
    replicated per exon, R/class_definition.R:119, :168), then per chromosome the padded chain of R/class_definition.R:364-368 through C_hmm --
    the same path and calls as the checker's CallCNVs re-enactment."""
    from exomedepth_amd import synth
    E, Cn = 30_000, 24
    chrom_off, start, end = synth.exon_design(E, Cn, 9)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, 2, 9, n_segments=12, mean_depth=80.0)
    t = 1e-4
    T = np.array([[1 - t, t / 2, t / 2], [.5, .5, 0], [.5, 0, .5]])
    L = 50000.0
    for s in range(2):
        tot = (test[:, s] + ref[:, s]).astype(np.int32)
        ll = edlib.get_loglike_matrix(np.full(E, phi[s]), np.full(E, p[s]), tot, test[:, s])
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], tot, test[:, s], 1.0, oracle.PORTABLE)
        assert np.array_equal(bits(ll), bits(ell))
        epath, ecalls = oracle.callcnvs(ell, chrom_off, start, end)
        path = np.zeros(E, dtype=np.int64)
        calls = []
        for c in range(Cn):
            lo, hi = int(chrom_off[c]), int(chrom_off[c + 1])
            if hi <= lo:
                continue
            # R/class_definition.R:364-368: a dummy first and last exon, HMM column order (normal, deletion, duplication)
            loc = np.vstack([[-np.inf, 0, -np.inf], ll[lo:hi], [-100, 0, -100]])[:, [1, 0, 2]]
            pos = np.concatenate([[start[lo] - 2 * L], start[lo:hi], [end[hi - 1] + 2 * L]]).astype(np.int32)
            res = edlib.viterbi_hmm(T, loc, pos, L)
            path[lo:hi] = res["Viterbi.path"][1:-1]
            for r in res["calls"]:
                calls.append((r["start.p"] - 1 + lo, r["end.p"] - 1 + lo, r["type"], r["nexons"]))      # :371-372, :409-410
        assert np.array_equal(path.astype(np.int8), epath)
        assert np.array_equal(np.array(calls, dtype=np.float64).reshape(-1, 4), ecalls)
