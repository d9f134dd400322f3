"""The R side of the drop-in, as source: shim/edcore_shim.c compiled against declarations-only R headers (tests/rapi/)
and driven through a miniature runtime (tests/rapi/mini_r.c) -- R itself is not in the image.

CPU part (no GPU): the shim compiles warning-free and links against libedcore.so; R_init_ExomeDepth registers exactly
what reference src/ExomeDepth_init.c:14-24 registers; the texts it prints / raises are the reference's
(src/CNV_estimate.cpp:61, src/hmm.cpp:38); without a device the .Call raises an R error (no CPU fallback).
GPU part: the two .Call entries through SEXPs give, bit for bit, what the ctypes path gives; the GSL error lines
(src/error.c:45-48) come out as the reference prints them.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAPI = os.path.join(ROOT, "tests", "rapi")
OUT = os.path.join(RAPI, "_build")

# typed in from the reference (src/CNV_estimate.cpp:61, src/hmm.cpp:38, src/error.c:45-48)
MIXTURE_FMT = "As a warning (this could be normal), the mixture coefficient is %f\n"
NSTATES_MSG = "ERROR: The code must assume 3 states"
HANDLER_LINE = "Default GSL error handler invoked.\n"


@pytest.fixture(scope="module")
def shim():
    from exomedepth_amd import _build
    if not os.path.exists(_build.LIB):
        _build.build()
    os.makedirs(OUT, exist_ok=True)
    minir = os.path.join(OUT, "libminir.so")
    so = os.path.join(OUT, "edcore_shim.so")
    libdir = os.path.dirname(_build.LIB)
    # tools/sanitize.sh: another compiler + sanitizer flags for the shim and the miniature runtime, the library's sanitizer build behind them
    cc = os.environ.get("ED_SHIM_CC", "gcc")
    extra = os.environ.get("ED_SHIM_CFLAGS", "").split()
    var = os.environ.get("ED_LIB_VARIANT")
    libflag = ("-l:libedcore_%s.so" % var) if var else "-ledcore"
    subprocess.run([cc, "-O1", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I", RAPI, os.path.join(RAPI, "mini_r.c"),
                    "-o", minir] + extra, check=True)
    # the shim exactly as an R package would build it (shim/Makevars), with tests/rapi standing where R's include dir is
    subprocess.run([cc, "-O2", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type", "-std=gnu99", "-shared", "-fPIC", "-I", RAPI,
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "shim", "edcore_shim.c"), "-o", so,
                    "-L", libdir, libflag, "-Wl,-rpath," + libdir] + extra, check=True)
    try:   # when torch shares the process its HIP runtime must come up first (see tests/conftest.py::edlib)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    R = C.CDLL(minir, mode=C.RTLD_GLOBAL)
    S = C.CDLL(so)
    vp = C.c_void_p
    for name, res, args in (("minir_n_registered", C.c_int, []), ("minir_registered_name", C.c_char_p, [C.c_int]),
                            ("minir_registered_nargs", C.c_int, [C.c_int]), ("minir_registered_fun", vp, [C.c_int]),
                            ("minir_dynamic_symbols", C.c_int, []), ("minir_output", C.c_char_p, []),
                            ("minir_reset_output", None, []), ("minir_error", C.c_char_p, []),
                            ("minir_protect_balance", C.c_int, []), ("minir_protect_max", C.c_int, []),
                            ("minir_real", vp, [vp, C.c_ssize_t]), ("minir_int", vp, [vp, C.c_ssize_t]),
                            ("minir_type", C.c_int, [vp]), ("minir_nrow", C.c_int, [vp]), ("minir_ncol", C.c_int, [vp]),
                            ("minir_call5", vp, [vp] * 6), ("minir_call6", vp, [vp] * 7), ("minir_callv", vp, [vp, C.c_int, C.POINTER(vp)]),
                            ("minir_int_matrix", vp, [vp, C.c_int, C.c_int]), ("minir_nil", vp, []), ("minir_name", C.c_char_p, [vp, C.c_int]),
                            ("minir_raw", C.POINTER(C.c_ubyte), [vp]), ("minir_int_data", C.POINTER(C.c_int), [vp]),
                            ("minir_fail_alloc_after", None, [C.c_int]), ("minir_run_finalizers", C.c_int, []),
                            ("REAL", C.POINTER(C.c_double), [vp]), ("XLENGTH", C.c_ssize_t, [vp]), ("VECTOR_ELT", vp, [vp, C.c_ssize_t])):
        fn = getattr(R, name)
        fn.restype, fn.argtypes = res, args
    S.R_init_ExomeDepth.argtypes = [vp]
    S.R_init_ExomeDepth.restype = None
    S.R_init_ExomeDepth(None)
    entries = {R.minir_registered_name(i).decode(): (R.minir_registered_nargs(i), R.minir_registered_fun(i))
               for i in range(max(R.minir_n_registered(), 0))}

    class Shim:
        pass
    sh = Shim()
    sh.R, sh.S, sh.entries = R, S, entries

    def real(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return R.minir_real(a.ctypes.data, a.size)

    def integer(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return R.minir_int(a.ctypes.data, a.size)

    def values(sexp):
        n = R.XLENGTH(sexp)
        return np.ctypeslib.as_array(R.REAL(sexp), shape=(n,)).copy() if n else np.zeros(0)

    def dot_call(name, *args):          # .Call(name, ...): by registered name, arity checked as R checks it
        nargs, fn = entries[name]
        assert len(args) == nargs
        R.minir_reset_output()
        arr = (vp * nargs)(*args)
        r = R.minir_callv(fn, nargs, arr)
        return r, R.minir_output().decode(), R.minir_error().decode()

    def int_matrix(a):                  # an R integer matrix: column-major
        a = np.asarray(a, dtype=np.int32)
        f = np.asfortranarray(a)
        return R.minir_int_matrix(f.ctypes.data, a.shape[0], a.shape[1])

    def as_list(sexp):                  # a named VECSXP -> dict of numpy arrays (REALSXP / INTSXP / RAWSXP / NULL)
        out = {}
        for i in range(R.XLENGTH(sexp)):
            el = R.VECTOR_ELT(sexp, i)
            name = R.minir_name(sexp, i).decode()
            t, n = R.minir_type(el), R.XLENGTH(el)
            if t == 0:
                out[name] = None
            elif t == 14:
                out[name] = values(el)
            elif t == 13:
                out[name] = np.ctypeslib.as_array(R.minir_int_data(el), shape=(n,)).copy() if n else np.zeros(0, np.int32)
                if R.minir_ncol(el) > 1 and R.minir_nrow(el) * R.minir_ncol(el) == n:
                    out[name] = out[name].reshape(R.minir_ncol(el), R.minir_nrow(el)).T      # column-major matrix
            elif t == 24:
                out[name] = np.ctypeslib.as_array(R.minir_raw(el), shape=(n,)).copy().reshape(R.minir_ncol(el), R.minir_nrow(el)).T
            else:
                raise AssertionError("unexpected SEXP type %d" % t)
        return out
    sh.real, sh.integer, sh.values, sh.dot_call, sh.int_matrix, sh.as_list, sh.nil = real, integer, values, dot_call, int_matrix, as_list, R.minir_nil()
    return sh


def test_registration_is_the_references(shim):
    """reference src/ExomeDepth_init.c:14-24: {"C_hmm", 6}, {"get_loglike_matrix", 5} first and unchanged, dynamic symbols off;
    next to them the three cohort-level entries of this library."""
    names = [shim.R.minir_registered_name(i).decode() for i in range(shim.R.minir_n_registered())]
    assert names[:2] == ["C_hmm", "get_loglike_matrix"]
    assert {k: v[0] for k, v in shim.entries.items()} == {"C_hmm": 6, "get_loglike_matrix": 5, "ed_call_cnvs_batch": 16,
                                                          "ed_fit_betabin_batch": 3, "ed_select_reference_set": 4, "ed_cohort_reference_sets": 4}
    assert shim.R.minir_dynamic_symbols() == 0
    for name in ("C_hmm", "get_loglike_matrix"):
        assert shim.entries[name][1] == C.cast(getattr(shim.S, name), C.c_void_p).value     # the registered pointers are the exported entries
    for name, cname in (("ed_call_cnvs_batch", "edr_call_cnvs_batch"), ("ed_fit_betabin_batch", "edr_fit_betabin_batch"),
                        ("ed_select_reference_set", "edr_select_reference_set"), ("ed_cohort_reference_sets", "edr_cohort_reference_sets")):
        assert shim.entries[name][1] == C.cast(getattr(shim.S, cname), C.c_void_p).value


def test_cohort_entries_check_their_arguments(shim):
    """shape errors are R errors with a message, before any device work"""
    t = shim.int_matrix(np.ones((5, 3))); r = shim.int_matrix(np.ones((5, 2)))
    res, out, err = shim.dot_call("ed_fit_betabin_batch", t, r, shim.integer([0]))
    assert res is None and "same shape" in err
    res, out, err = shim.dot_call("ed_select_reference_set", shim.integer([1, 2, 3]), t, shim.nil, shim.integer([0]))
    assert res is None and err.startswith("The number of rows of the reference matrix must match")


def test_shim_texts_are_the_references(shim):
    src = open(os.path.join(ROOT, "shim", "edcore_shim.c")).read()
    assert '"' + MIXTURE_FMT.replace("\n", "\\n") + '"' in src
    assert '"' + NSTATES_MSG + '"' in src


def test_mixture_notice_and_no_cpu_fallback(shim):
    n = 4
    r, out, err = shim.dot_call("get_loglike_matrix", shim.real(np.full(n, 0.01)), shim.real(np.full(n, 0.2)),
                                shim.integer([100, 50, 0, 10]), shim.integer([20, 5, 0, 3]), shim.real([0.5]))
    assert out.startswith(MIXTURE_FMT % 0.5)
    from exomedepth_amd import _lib
    if _lib.lib().ed_device_count() == 0:
        assert r is None and "no usable HIP device" in err          # an R error, not a silently computed matrix
    else:
        assert r is not None and err == "" and (shim.R.minir_nrow(r), shim.R.minir_ncol(r)) == (n, 3)
    assert shim.R.minir_protect_balance() == 0 or r is None


def test_nstates_other_than_three_raises(shim):
    r, out, err = shim.dot_call("C_hmm", shim.integer([2]), shim.integer([5]), shim.real(np.full(4, 0.5)), shim.real(np.zeros(10)),
                                shim.integer(np.arange(5)), shim.real([1.0]))
    assert r is None and err == NSTATES_MSG


@pytest.mark.gpu
def test_get_loglike_matrix_through_sexp_equals_ctypes_path(shim, edlib):
    rng = np.random.default_rng(5)
    n = 5000
    phi = rng.uniform(0.001, 0.3, n); e = rng.uniform(0.02, 0.9, n)
    tot = rng.integers(0, 3000, n).astype(np.int32); obs = (tot * rng.uniform(0, 1, n)).astype(np.int32)
    for mix in (1.0, 0.4):
        r, out, err = shim.dot_call("get_loglike_matrix", shim.real(phi), shim.real(e), shim.integer(tot), shim.integer(obs),
                                    shim.real([mix]))
        assert r is not None and err == ""
        assert out == ("" if mix == 1.0 else MIXTURE_FMT % mix)
        assert (shim.R.minir_type(r), shim.R.minir_nrow(r), shim.R.minir_ncol(r)) == (14, n, 3)     # REALSXP n x 3
        got = shim.values(r).reshape(3, n).T                                                        # column-major
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            want = edlib.get_loglike_matrix(phi, e, tot, obs, mixture=mix)
        assert np.array_equal(np.ascontiguousarray(got).view(np.int64), np.ascontiguousarray(want).view(np.int64))
        assert shim.R.minir_protect_balance() == 0


@pytest.mark.gpu
def test_c_hmm_through_sexp_equals_ctypes_path(shim, edlib):
    T = np.full((3, 3), 1 / 3)
    ll = np.array([[0, -10, -10]] * 3 + [[-10, -10, 0]] * 3 + [[-10, 0, -10]] * 4, dtype=float)   # reference R/tools.R:74-79
    pos = np.arange(1, 11)
    cases = [(T, ll, pos, 1.0)]
    rng = np.random.default_rng(9)
    t = 1e-3
    for n in (2, 57, 4000):
        cases.append((np.array([[1 - t, t / 2, t / 2], [.5, .5, 0], [.5, 0, .5]]), rng.normal(-3, 3, (n, 3)),
                      np.cumsum(rng.integers(1, 30000, n)), 50000.0))
    for Tm, llm, p, L in cases:
        n = llm.shape[0]
        r, out, err = shim.dot_call("C_hmm", shim.integer([3]), shim.integer([n]), shim.real(Tm.T.ravel()), shim.real(llm.T.ravel()),
                                    shim.integer(p), shim.real([L]))
        assert r is not None and err == "" and out == ""
        assert shim.R.minir_type(r) == 19 and shim.R.XLENGTH(r) == 2                              # VECSXP of 2 (src/hmm.cpp:133)
        path, calls = shim.R.VECTOR_ELT(r, 0), shim.R.VECTOR_ELT(r, 1)
        want = edlib.viterbi_hmm(Tm, llm, p, L)
        assert shim.R.minir_type(path) == 14 and np.array_equal(shim.values(path), want["Viterbi.path"].astype(float))
        k = len(want["calls"])
        assert (shim.R.minir_type(calls), shim.R.minir_nrow(calls), shim.R.minir_ncol(calls)) == (14, k, 4)
        got = shim.values(calls).reshape(4, k)
        for j, name in enumerate(("start.p", "end.p", "type", "nexons")):
            assert np.array_equal(got[j], want["calls"][name])
        assert shim.R.minir_protect_balance() == 0
    path, calls = shim.R.VECTOR_ELT(shim.dot_call("C_hmm", shim.integer([3]), shim.integer([10]), shim.real(T.T.ravel()),
                                                  shim.real(ll.T.ravel()), shim.integer(pos), shim.real([1.0]))[0], 0), None
    assert shim.values(path).tolist() == [0, 0, 0, 2, 2, 2, 1, 1, 1, 0]


@pytest.mark.gpu
def test_gsl_error_lines_are_printed_as_the_reference_prints_them(shim, edlib):
    """expected = 0 makes every shape parameter NaN: each of the row's six gsl_sf_lnbeta calls fails three times in
    gsl_sf_lngamma_sgn_e (src/VP_gamma.c:1283, GSL_EROUND) and once in the natural-prototype wrapper (src/beta.c:163),
    two lines per gsl_error call (src/error.c:45-48); the values are 0.0.  total = observed = 0 with a zero
    shape parameter hits the x == 0 domain error of src/beta.c:56."""
    phi = np.array([0.01, 0.01, 0.01]); e = np.array([0.2, 0.0, 0.2])
    tot = np.array([100, 10, 7], np.int32); obs = np.array([20, 3, 1], np.int32)
    r, out, err = shim.dot_call("get_loglike_matrix", shim.real(phi), shim.real(e), shim.integer(tot), shim.integer(obs), shim.real([1.0]))
    assert r is not None and err == ""
    got = shim.values(r).reshape(3, 3).T
    assert np.all(got[1] == 0.0) and np.all(np.isfinite(got))
    one_call = ("ERROR VP_gamma.c 1283 error\n" + HANDLER_LINE) * 3 + "ERROR beta.c 163 gsl_sf_lnbeta_e(x, y, &result)\n" + HANDLER_LINE
    assert out == one_call * 6


# ---- the cohort-level entries through SEXPs ----------------------------------------------------------------------
def _cohort_case(seed=21, E=9000, C_=4, S=150):
    from exomedepth_amd import synth
    chrom_off, start, end = synth.exon_design(E, C_, seed=seed)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed=seed, n_segments=4, mean_depth=70.0)
    return chrom_off, start, end, test, ref, p, phi


@pytest.mark.gpu
@pytest.mark.parametrize("given,slab,mode,emit", [(False, 64, 0, 0), (True, 150, 0, 0), (False, 150, 1, 0), (True, 150, 0, 2), (False, 64, 0, 2), (True, 64, 0, 1)])
def test_call_cnvs_batch_through_sexp_equals_ctypes_path(shim, edlib, given, slab, mode, emit):
    """.Call("ed_call_cnvs_batch", ...) on R's column-major integer matrices = the batch interface on the same data, bit for bit:
    call table, decoration (R/class_definition.R:379-405), fitted parameters, Viterbi path."""
    chrom_off, start, end, test, ref, p, phi = _cohort_case()
    E, S = test.shape
    res, out, err = shim.dot_call("ed_call_cnvs_batch", shim.int_matrix(test), shim.int_matrix(ref), shim.integer(chrom_off), shim.integer(start),
                                  shim.integer(end), shim.real([1e-4]), shim.real([50000.0]), shim.real(phi) if given else shim.nil,
                                  shim.real(p) if given else shim.nil, shim.real([1.0]), shim.integer([slab]), shim.integer([1]), shim.integer([mode]),
                                  shim.integer([1]), shim.integer([emit]), shim.nil)
    assert res is not None and err == "" and out == ""
    got = shim.as_list(res)
    assert list(got) == ["sample", "start.p", "end.p", "type", "nexons", "BF", "reads.expected", "reads.observed", "reads.ratio", "phi",
                         "expected", "path", "n.unconverged", "n.gsl.errors", "phi.bins", "complete.bins"]
    assert got["phi.bins"] is None and got["complete.bins"] is None
    plan = edlib.Plan(chrom_off, start, end)
    b = edlib.Batch(plan, S)
    if emit:                 # (emit.mode: the same table-driven arithmetic through the ctypes path; the fit is the same in every mode)
        b.set_emit_mode(emit)
    if mode:
        from exomedepth_amd._lib import check, lib
        check(lib().ed_batch_set_fit_mode(b.handle, 1))
    if given:
        dphi, dexp = edlib.DeviceArray(phi), edlib.DeviceArray(p)
    else:
        dphi, dexp = edlib.DeviceArray(np.zeros(S)), edlib.DeviceArray(np.zeros(S))
        if slab < S:      # the fit is per sample, but its start values are shared work of a slab: fit slab by slab as the pipeline does
            fp, fe = np.zeros(S), np.zeros(S)
            for s0 in range(0, S, slab):
                n = min(slab, S - s0)
                bb = edlib.Batch(plan, n)
                a, c = edlib.DeviceArray(np.zeros(n)), edlib.DeviceArray(np.zeros(n))
                bb.fit(np.ascontiguousarray(test[:, s0:s0 + n]), np.ascontiguousarray(ref[:, s0:s0 + n]), a, c)
                edlib.api.check(edlib.api.lib().ed_synchronize(None))
                fp[s0:s0 + n], fe[s0:s0 + n] = a.to_host(), c.to_host()
                bb.close()
            dphi, dexp = edlib.DeviceArray(fp), edlib.DeviceArray(fe)
        else:
            b.fit(test, ref, dphi, dexp)
    b.run(test, ref, dphi, dexp)
    calls, info = b.calls(), b.call_info()
    assert np.array_equal(got["sample"], calls["sample"] + 1) and np.array_equal(got["start.p"], calls["start_exon"] + 1)
    assert np.array_equal(got["end.p"], calls["end_exon"] + 1) and np.array_equal(got["type"], calls["type"])
    assert np.array_equal(got["nexons"], calls["nexons"]) and len(calls) > 50
    assert got["BF"].tobytes() == info["BF"].tobytes() and got["reads.ratio"].tobytes() == info["reads_ratio"].tobytes()
    assert np.array_equal(got["reads.expected"], info["reads_expected"]) and np.array_equal(got["reads.observed"], info["reads_observed"].astype(float))
    if emit == 2 and not given:   # the sample-major fit adds the starting moments up in another order: the same estimate to the fit's tolerance
        assert np.all(np.abs(got["phi"] - dphi.to_host()) <= 1e-9 * dphi.to_host())
    else:
        assert got["phi"].tobytes() == dphi.to_host().tobytes() and got["expected"].tobytes() == dexp.to_host().tobytes()
    assert np.array_equal(got["path"], b.path())                       # raw n_exons x n_samples
    assert got["n.unconverged"][0] == 0 and got["n.gsl.errors"][0] == 0
    assert shim.R.minir_protect_balance() == 0
    b.close(); plan.close()


@pytest.mark.gpu
def test_call_cnvs_batch_leaves_nothing_behind_when_r_unwinds(shim, edlib):
    """an R error inside one of the entry's allocations (allocation failure: a longjmp out of the .Call) leaves the plan and the cohort
    to their external pointer's finalizer -- nothing leaks; the entries that return normally have released theirs already"""
    chrom_off, start, end, test, ref, p, phi = _cohort_case(E=3000, S=40)
    args = lambda: (shim.int_matrix(test), shim.int_matrix(ref), shim.integer(chrom_off), shim.integer(start), shim.integer(end), shim.real([1e-4]),
                    shim.real([50000.0]), shim.nil, shim.nil, shim.real([1.0]), shim.integer([40]), shim.integer([1]), shim.integer([0]),
                    shim.integer([1]), shim.integer([0]), shim.nil)
    shim.R.minir_run_finalizers()
    res, out, err = shim.dot_call("ed_call_cnvs_batch", *args())
    assert res is not None and err == ""
    assert shim.R.minir_run_finalizers() == 0                     # a normal return: released by the call itself
    for k in (2, 3, 4, 5, 8):                                     # the k-th allocation of the call fails (1 is the guard itself: nothing made yet)
        a = args()
        shim.R.minir_fail_alloc_after(k)
        res, out, err = shim.dot_call("ed_call_cnvs_batch", *a)
        shim.R.minir_fail_alloc_after(-1)
        assert res is None and "cannot allocate" in err
        assert shim.R.minir_run_finalizers() == 1                 # the collector's turn: the objects the unwinding left behind
        assert shim.R.minir_run_finalizers() == 0
    res, out, err = shim.dot_call("ed_call_cnvs_batch", *args())  # and the library is as it was
    assert res is not None and err == ""


@pytest.mark.gpu
@pytest.mark.parametrize("emit,B", [(2, 1), (0, 1), (0, 3)])
def test_call_cnvs_batch_over_several_devices(shim, edlib, emit, B):
    """.Call("ed_call_cnvs_batch", ..., devices = c(0L, 0L)): the cohort's columns dealt to two pipelines, each driven by its own host thread
    (on this box both on device 0; on a node `devices = NULL` takes every visible GPU), call tables merged in column order -- the very
    list the single-device call returns, bit for bit (same slabs: the fit's start values are shared work of a slab)."""
    chrom_off, start, end, test, ref, p, phi = _cohort_case(S=200)
    E, S = test.shape
    slab = 50                        # 4 slabs of the 200 columns: 2 per device
    assert S == 4 * slab
    def call(devs):
        res, out, err = shim.dot_call("ed_call_cnvs_batch", shim.int_matrix(test), shim.int_matrix(ref), shim.integer(chrom_off), shim.integer(start),
                                      shim.integer(end), shim.real([1e-4]), shim.real([50000.0]), shim.nil, shim.nil, shim.real([1.0]),
                                      shim.integer([slab]), shim.integer([1]), shim.integer([0]), shim.integer([B]), shim.integer([emit]),
                                      shim.integer(devs) if devs is not None else shim.nil)
        assert res is not None and err == "" and out == "", err
        return shim.as_list(res)
    one, two, three = call([0]), call([0, 0]), call([0, 0, 0])
    for got in (two, three):
        for k in one:
            a, b = one[k], got[k]
            if a is None:
                assert b is None, k
            else:
                assert np.asarray(a).tobytes() == np.asarray(b).tobytes(), k
    assert len(one["sample"]) > 50 and np.all(np.diff(one["sample"]) >= 0)
    assert shim.R.minir_protect_balance() == 0
    # a device that does not exist is an R error with the library's message
    res, out, err = shim.dot_call("ed_call_cnvs_batch", shim.int_matrix(test), shim.int_matrix(ref), shim.integer(chrom_off), shim.integer(start),
                                  shim.integer(end), shim.real([1e-4]), shim.real([50000.0]), shim.nil, shim.nil, shim.real([1.0]),
                                  shim.integer([slab]), shim.integer([0]), shim.integer([0]), shim.integer([B]), shim.integer([emit]), shim.integer([0, 99]))
    assert res is None and "device 99" in err          # (an R error unwinds the protect stack itself)
    assert shim.R.minir_run_finalizers() <= 1


@pytest.mark.gpu
def test_call_cnvs_batch_with_phi_bins_through_sexp(shim, edlib):
    """the reference's phi.bins argument (R/class_definition.R:86, :120-147) through .Call("ed_call_cnvs_batch", ..., phi.bins = 3):
    phi.estimates per depth level and complete.bins come back as matrices, `phi` is NA, and calls / decoration / path are those of
    ed_batch_run_bins given these parameters, bit for bit"""
    chrom_off, start, end, test, ref, p, phi = _cohort_case()
    E, S = test.shape
    B = 3
    res, out, err = shim.dot_call("ed_call_cnvs_batch", shim.int_matrix(test), shim.int_matrix(ref), shim.integer(chrom_off), shim.integer(start),
                                  shim.integer(end), shim.real([1e-4]), shim.real([50000.0]), shim.nil, shim.nil, shim.real([1.0]),
                                  shim.integer([150]), shim.integer([1]), shim.integer([0]), shim.integer([B]), shim.integer([0]), shim.nil)
    assert res is not None and err == "" and out == ""
    got = shim.as_list(res)
    phib = np.asarray(got["phi.bins"]).reshape(S, B).T            # B x S, column-major
    edges = np.asarray(got["complete.bins"]).reshape(S, B + 1).T
    assert np.all(np.isnan(got["phi"])) and np.all(phib > 0) and np.all(np.diff(edges, axis=0) >= 0)
    plan = edlib.Plan(chrom_off, start, end)
    b = edlib.Batch(plan, S)
    d = [edlib.DeviceArray(np.zeros((B, S))), edlib.DeviceArray(np.zeros((B + 1, S))), edlib.DeviceArray(np.zeros(S))]
    b.fit_bins(test, ref, B, *d)
    assert np.array_equal(edges, d[1].to_host()) and np.max(np.abs(phib - d[0].to_host()) / phib) < 1e-9
    b.run_bins(test, ref, B, np.ascontiguousarray(phib), np.ascontiguousarray(edges), got["expected"])
    calls, info = b.calls(), b.call_info()
    assert np.array_equal(got["sample"], calls["sample"] + 1) and np.array_equal(got["start.p"], calls["start_exon"] + 1)
    assert np.array_equal(got["end.p"], calls["end_exon"] + 1) and np.array_equal(got["type"], calls["type"]) and len(calls) > 50
    assert got["BF"].tobytes() == info["BF"].tobytes() and np.array_equal(got["path"], b.path())
    b.close(); plan.close()
    # parameters cannot be given in this mode: an R error before any device work
    res, out, err = shim.dot_call("ed_call_cnvs_batch", shim.int_matrix(test), shim.int_matrix(ref), shim.integer(chrom_off), shim.integer(start),
                                  shim.integer(end), shim.real([1e-4]), shim.real([50000.0]), shim.real(phi), shim.real(p), shim.real([1.0]),
                                  shim.integer([150]), shim.integer([0]), shim.integer([0]), shim.integer([B]), shim.integer([0]), shim.nil)
    assert res is None and "phi.bins" in err
    assert shim.R.minir_protect_balance() == 0


@pytest.mark.gpu
def test_fit_betabin_batch_through_sexp(shim, edlib, oracle):
    """.Call("ed_fit_betabin_batch", test, reference, mode): what stands where R/class_definition.R:118 calls aod::betabin.  Mode 0
    = the maximum-likelihood estimate (against the checker's long-double MLE); mode 1 = aod's own procedure (against the checker's
    statement-by-statement restatement of R's nmmin on the same objective)."""
    chrom_off, start, end, test, ref, p, phi = _cohort_case(seed=33, E=6000, S=40)
    E, S = test.shape
    for mode in (0, 1):
        res, out, err = shim.dot_call("ed_fit_betabin_batch", shim.int_matrix(test), shim.int_matrix(ref), shim.integer([mode]))
        assert res is not None and err == ""
        got = shim.as_list(res)
        assert list(got) == ["phi", "expected", "converged"] and np.all(got["converged"] == 1)
        for s in (0, 7, 39):
            if mode == 0:
                ophi, op, _, _ = oracle.fit_mle(test[:, s], ref[:, s])
                assert abs(got["phi"][s] - ophi) < 1e-8 * ophi and abs(got["expected"][s] - op) < 1e-8 * op
            else:
                ophi, op, ne, fail = oracle.fit_nm(test[:, s], ref[:, s], with_status=True)
                assert fail == 0
                # two evaluations of the same objective that differ in the last bits can part ways at a comparison of the search;
                # both then stop inside optim()'s tolerance region: 1e-3 relative in phi is that region's size
                assert abs(got["phi"][s] - ophi) < 2e-3 * ophi and abs(got["expected"][s] - op) < 2e-4 * op
    assert shim.R.minir_protect_balance() == 0


@pytest.mark.gpu
def test_select_reference_set_through_sexp(shim, edlib):
    """.Call("ed_select_reference_set", test.counts, reference.counts, bin.length, n.bins.reduced) = api.select_reference_set"""
    rng = np.random.default_rng(4)
    E, R = 4000, 12
    lam = rng.lognormal(np.log(90), 0.7, E)
    noise = np.linspace(0.02, 0.5, R)
    refs = rng.poisson(lam[:, None] * np.exp(rng.normal(0, noise[None, :], (E, R)))).astype(np.int32)
    test = rng.poisson(lam).astype(np.int32)
    bl = rng.integers(80, 400, E).astype(float)
    want = edlib.select_reference_set(test, refs, bl, 0)
    res, out, err = shim.dot_call("ed_select_reference_set", shim.integer(test), shim.int_matrix(refs), shim.real(bl), shim.integer([0]))
    assert res is not None and err == ""
    got = shim.as_list(res)
    st = want["summary.stats"]
    assert np.array_equal(got["ref.samples"], st["ref_index"] + 1)
    assert [("X%d" % i) for i in got["reference.choice"]] == want["reference.choice"]
    for a, b in (("correlations", "correlation"), ("expected.BF", "expected_BF"), ("phi", "phi"), ("RatioSd", "ratio_sd"), ("mean.p", "mean_p"),
                 ("median.depth", "median_depth")):
        assert got[a].tobytes() == st[b].tobytes(), a
    assert np.array_equal(got["selected"], st["selected"]) and got["n.bins"][0] == want["n.bins"]
    assert shim.R.minir_protect_balance() == 0


@pytest.mark.gpu
def test_cohort_reference_sets_through_sexp(shim, edlib):
    """.Call("ed_cohort_reference_sets", counts, bin.length, n.bins.reduced, max.refs) = api.cohort_select_reference_sets on the same
    cohort: choices (1-based, NA padded), the aggregate reference matrix in R's layout"""
    rng = np.random.default_rng(12)
    E, S = 6000, 20
    lam = rng.lognormal(np.log(90), 0.7, E)
    counts = rng.poisson(lam[:, None] * rng.lognormal(0, 0.2, S)[None, :] * np.exp(rng.normal(0, 0.1, (E, S)))).astype(np.int32)
    bl = rng.integers(80, 400, E).astype(float)
    want = edlib.cohort_select_reference_sets(counts, bl, 0, max_refs=19)
    res, out, err = shim.dot_call("ed_cohort_reference_sets", shim.int_matrix(counts), shim.real(bl), shim.integer([0]), shim.integer([19]))
    assert res is not None and err == ""
    got = shim.as_list(res)
    assert list(got) == ["n.chosen", "choice", "reference", "correlations", "n.bins"]
    assert np.array_equal(got["n.chosen"], want["n_chosen"]) and got["n.bins"][0] == want["n.bins"]
    NA = -2147483648
    for t in range(S):
        k = want["n_chosen"][t]
        assert np.array_equal(got["choice"][:k, t], want["choice"][t, :k] + 1) and np.all(got["choice"][k:, t] == NA)
    assert np.array_equal(got["reference"], want["reference"].to_host().reshape(E, S))
    assert shim.R.minir_protect_balance() == 0


def _r_dot_calls(text):
    """every .Call("name", args...) of an R source text: (name, number of arguments before PACKAGE =)"""
    import re
    text = "\n".join(line.split("#", 1)[0] for line in text.splitlines()) + "\n"     # R comments (no '#' inside a string in these sources)
    out = []
    for m in re.finditer(r'\.Call\(\s*"([A-Za-z0-9_.]+)"', text):
        i, depth, args, cur = m.end(), 1, [], ""
        while depth > 0:
            c = text[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if c == "," and depth == 1:
                args.append(cur.strip()); cur = ""
            elif c == "#":                       # an R comment runs to the end of the line
                while text[i] != "\n":
                    i += 1
                continue
            else:
                cur += c
            i += 1
        args.append(cur.strip())
        args = [a for a in args if a and not a.startswith("PACKAGE")]
        out.append((m.group(1), len(args)))
    return out


def test_r_wrappers_match_the_registered_entries(shim):
    """shim/R/exomedepth_amd.R (R cannot run here): every .Call names an entry the compiled shim registers and passes as many arguments
    as that entry's registered arity -- what R itself checks at call time (reference src/ExomeDepth_init.c:14-24); and the reference's
    own two call sites (R/class_definition.R:184-189, R/tools.R:97) still fit the first two rows of the table"""
    import os
    text = open(os.path.join(os.path.dirname(__file__), "..", "shim", "R", "exomedepth_amd.R")).read()
    calls = _r_dot_calls(text)
    assert len(calls) >= 5
    arity = {k: v[0] for k, v in shim.entries.items()}
    for name, nargs in calls:
        assert name in arity, name
        assert nargs == arity[name], (name, nargs, arity[name])
    assert {n for n, _ in calls} == {"ed_call_cnvs_batch", "ed_fit_betabin_batch", "ed_select_reference_set", "ed_cohort_reference_sets"}
    # the reference's call sites, as its R sources write them
    ref_sites = '.Call("get_loglike_matrix", phi = a, expected = b, total = as.integer(c), observed = as.integer(d), mixture = e, PACKAGE = "ExomeDepth")\n' \
                '.Call("C_hmm", nrow(T), nrow(ll), T, ll, positions, as.double(L), PACKAGE = "ExomeDepth")'
    assert [(n, k) for n, k in _r_dot_calls(ref_sites)] == [("get_loglike_matrix", 5), ("C_hmm", 6)]
    assert arity["get_loglike_matrix"] == 5 and arity["C_hmm"] == 6
