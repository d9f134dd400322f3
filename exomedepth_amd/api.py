"""Host-side mirror of the reference's interface for the hot path, on top of the C-ABI (libedcore.so).

Names, argument meaning and error behaviour follow the reference's R layer so that the parity tests
read like the reference's own examples:

    get_loglike_matrix(...)   <- .Call("get_loglike_matrix", ...)      reference R/class_definition.R:184-189
    viterbi_hmm(...)          <- viterbi.hmm()                         reference R/tools.R:88-103
    ExomeDepth(...)           <- new('ExomeDepth', test=, reference=)  reference R/class_definition.R:82-191
    ExomeDepth.CallCNVs(...)  <- CallCNVs()                            reference R/class_definition.R:311-419
    ExomeDepth.TestCNV(...)   <- TestCNV()                             reference R/class_definition.R:243-256
    Plan / Batch              the batched, device-resident interface (no counterpart in the reference,
                              whose granularity is one sample and one chromosome per call)

All arithmetic of the path runs on the GPU; this module only marshals buffers (numpy on the host,
optionally torch CUDA tensors for device-resident inputs).
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import EdCall, EdCallInfo, EdError, EdRefsetRow, check, lib

CALL_DTYPE = np.dtype([("sample", "<i4"), ("chrom", "<i4"), ("start_exon", "<i4"), ("end_exon", "<i4"),
                       ("type", "<i4"), ("nexons", "<i4")])
assert CALL_DTYPE.itemsize == C.sizeof(EdCall)
CALL_INFO_DTYPE = np.dtype([("BF_raw", "<f8"), ("BF", "<f8"), ("reads_expected", "<i8"), ("reads_observed", "<i8"),
                            ("reads_ratio", "<f8")])
assert CALL_INFO_DTYPE.itemsize == C.sizeof(EdCallInfo)


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _as_r_integer(x):
    """R's as.integer() on a numeric vector: truncation toward zero (R/class_definition.R:187-188)."""
    a = np.asarray(x)
    if a.dtype.kind in "iu":
        return a.astype(np.int32)
    return np.trunc(a).astype(np.int32)


# ---------------------------------------------------------------------------------------------
# the two .Call drop-ins
# ---------------------------------------------------------------------------------------------
def get_loglike_matrix(phi, expected, total, observed, mixture=1.0, return_errors=False):
    """Per-exon log-likelihoods of the three copy-number states; (n,3) array with columns
    (deletion, normal, duplication) -- reference src/CNV_estimate.cpp:52-85."""
    total = _i32(total)
    n = total.size
    phi = _f64(np.broadcast_to(np.asarray(phi, dtype=np.float64), (n,)))
    expected = _f64(np.broadcast_to(np.asarray(expected, dtype=np.float64), (n,)))
    observed = _i32(observed)
    if observed.size != n:
        raise ValueError("total and observed must have the same length")
    if mixture != 1:
        # reference src/CNV_estimate.cpp:61
        sys.stdout.write("As a warning (this could be normal), the mixture coefficient is %f\n" % mixture)
    out = np.empty((3, n), dtype=np.float64)  # column-major n x 3
    nerr = C.c_int64(0)
    check(lib().ed_get_loglike_matrix(_ptr(phi), _ptr(expected), _ptr(total), _ptr(observed), n, float(mixture),
                                      _ptr(out), C.byref(nerr)))
    return (out.T, nerr.value) if return_errors else out.T


def device_count():
    """HIP devices visible to the process (ed_device_count)"""
    return int(lib().ed_device_count())


def viterbi_hmm(transitions, loglikelihood, positions, expected_CNV_length):
    """reference R/tools.R:88-103.  loglikelihood: (nobs, nstates) in HMM order (normal, deletion,
    duplication).  Returns {'Viterbi.path': int array, 'calls': structured array with fields
    start.p, end.p, type, nexons} (1-based positions like the reference)."""
    T = np.asarray(transitions, dtype=np.float64)
    ll = np.asarray(loglikelihood, dtype=np.float64)
    if T.ndim != 2 or T.shape[0] != T.shape[1]:
        raise ValueError("Transition matrix is not square")
    positions = _i32(positions)
    if positions.size != ll.shape[0]:
        raise ValueError("The number of positions are not matching the number of rows of the likelihood matrix "
                         "%d and %d" % (positions.size, ll.shape[0]))
    nstates, nobs = T.shape[0], ll.shape[0]
    Tc = _f64(T.T.ravel())
    llc = _f64(ll.T.ravel())
    path = np.empty(nobs, dtype=np.float64)
    cap = max(nobs, 1)
    calls = np.zeros((4, cap), dtype=np.float64)  # column-major cap x 4
    nc = C.c_int64(0)
    check(lib().ed_hmm(nstates, nobs, _ptr(Tc), _ptr(llc), _ptr(positions), float(expected_CNV_length), _ptr(path),
                       _ptr(calls), cap, C.byref(nc)))
    k = nc.value
    rec = np.zeros(k, dtype=[("start.p", "f8"), ("end.p", "f8"), ("type", "f8"), ("nexons", "f8")])
    for j, name in enumerate(rec.dtype.names):
        rec[name] = calls[j, :k]
    return {"Viterbi.path": path.astype(np.int64), "calls": rec}


# ---------------------------------------------------------------------------------------------
# device buffers
# ---------------------------------------------------------------------------------------------
class DeviceArray:
    """A device allocation made through the library (for callers without torch)."""

    def __init__(self, host=None, nbytes=None):
        self.ptr = C.c_void_p()
        self.host_dtype = None
        self.shape = None
        if host is not None:
            host = np.ascontiguousarray(host)
            nbytes = host.nbytes
            self.host_dtype, self.shape = host.dtype, host.shape
        self.nbytes = int(nbytes)
        check(lib().ed_malloc(C.byref(self.ptr), self.nbytes))
        if host is not None and self.nbytes:
            check(lib().ed_memcpy_h2d(self.ptr, _ptr(host), self.nbytes))

    def to_host(self, dtype=None, shape=None):
        out = np.empty(shape if shape is not None else self.shape, dtype=dtype if dtype is not None else self.host_dtype)
        if out.nbytes:
            check(lib().ed_memcpy_d2h(_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().ed_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _RawDevice:
    """a bare device pointer owned by somebody else (e.g. a cohort slot's fitted parameters)"""

    def __init__(self, ptr):
        self.ptr = C.c_void_p(ptr)


def _device_pointer(x, dtype, keep):
    """Return a c_void_p for x: a torch CUDA tensor (used in place), a DeviceArray, or host data
    (uploaded; the temporary is appended to `keep`)."""
    if isinstance(x, (DeviceArray, _RawDevice)):
        return x.ptr
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):
        if not x.is_cuda:
            x = x.cuda()
            keep.append(x)
        if not x.is_contiguous():
            raise ValueError("device tensors must be contiguous")
        want = {np.dtype(np.int32): ("torch.int32",), np.dtype(np.float64): ("torch.float64",),
                np.dtype(np.uint16): ("torch.uint16", "torch.int16")}[np.dtype(dtype)]     # (16-bit counts: the bit pattern is what counts)
        if str(x.dtype) not in want:
            raise ValueError("expected a %s tensor, got %s" % (want[0], x.dtype))
        return C.c_void_p(x.data_ptr())
    if np.dtype(dtype) == np.dtype(np.uint16):
        xa = np.asarray(x)
        if xa.dtype != np.uint16 and xa.size and (xa.min() < 0 or xa.max() > 65535):
            raise ValueError("16-bit counts (set_counts_bits(16) / counts_bits = 16) hold 0 .. 65535")
    d = DeviceArray(np.ascontiguousarray(x, dtype=dtype))
    keep.append(d)
    return d.ptr


class Plan:
    """Exon design + HMM parameters (CallCNVs arguments, reference R/class_definition.R:261, :311).
    Exons must be ordered by (chromosome, position); chrom_off delimits the chromosomes."""

    def __init__(self, chrom_off, start, end, transition_probability=1e-4, expected_CNV_length=50000.0, device=0):
        self.chrom_off = _i32(chrom_off)
        self.start = _i32(start)
        self.end = _i32(end)
        self.n_exons = int(self.start.size)
        self.n_chrom = int(self.chrom_off.size - 1)
        self.transition_probability = float(transition_probability)
        self.expected_CNV_length = float(expected_CNV_length)
        self.handle = C.c_void_p()
        check(lib().ed_plan_create(C.byref(self.handle), int(device), self.n_exons, self.n_chrom, _ptr(self.chrom_off),
                                   _ptr(self.start), _ptr(self.end), self.transition_probability,
                                   self.expected_CNV_length))

    def close(self):
        if self.handle:
            lib().ed_plan_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """Device working set for n_samples samples of a plan.  Count matrices are int32
    [n_exons][n_samples] (sample-minor)."""

    STAGES = ("sample_consts", "emissions", "viterbi", "call_table", "fit")

    def __init__(self, plan, n_samples):
        self.plan = plan
        self.n_samples = int(n_samples)
        self.handle = C.c_void_p()
        self._keep_run = []   # device temporaries uploaded for the last run*(): call_info() / verify read the run's inputs
        self._keep_fit = []   # ... for the last fit*() (outputs may be read by the caller after the call)
        check(lib().ed_batch_create(C.byref(self.handle), plan.handle, self.n_samples))

    _owned = True

    @classmethod
    def _view(cls, plan, n_samples, handle):
        """a Batch interface on a batch object owned by somebody else (a Cohort's slot): close() leaves it alone"""
        b = cls.__new__(cls)
        b.plan, b.n_samples, b.handle = plan, int(n_samples), C.c_void_p(handle)
        b._keep_run, b._keep_fit, b._owned = [], [], False
        return b

    def close(self):
        if self.handle and self._owned:
            lib().ed_batch_destroy(self.handle)
        self.handle = C.c_void_p()
        self._keep_run = []
        self._keep_fit = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enable_timing(self, on=True):
        check(lib().ed_batch_enable_timing(self.handle, 1 if on else 0))

    def set_fused(self, fused=True):
        """fused=True: emissions + Viterbi as one kernel (minimal HBM traffic); default is the two-kernel path."""
        check(lib().ed_batch_set_fused(self.handle, 1 if fused else 0))

    def keep_loglik(self, keep=True):
        """Keep (default) or drop the (n_exons, 3, n_samples) likelihood matrix -- 24 bytes per cell of HBM."""
        check(lib().ed_batch_keep_loglik(self.handle, 1 if keep else 0))

    def set_async_tail(self, on=True):
        """on: run() leaves the Viterbi tail and the call table on streams of the batch instead of joining them into the
        caller's stream (two batches used alternately then overlap; see ed_batch_set_async_tail)."""
        check(lib().ed_batch_set_async_tail(self.handle, 1 if on else 0))

    def set_viterbi_overlap(self, on=True):
        """on (default): chains of a chromosome group run underneath the emissions of the next groups; off: all emissions,
        then all chains (what a pipeline of batches wants, see ed_batch_set_viterbi_overlap)"""
        check(lib().ed_batch_set_viterbi_overlap(self.handle, 1 if on else 0))

    def wait(self, stream=None):
        """make `stream` wait (on the device) for the last run() of this batch, asynchronous tail included"""
        check(lib().ed_batch_wait(self.handle, C.c_void_p(stream or 0)))

    def set_fit_histograms(self, on=True):
        """fit(): iterate on per-sample count histograms (default; True / 1: geometry picked from the data's depth,
        8 / 4 / 2: that geometry) or per cell on every pass (False / 0)."""
        check(lib().ed_batch_set_fit_histograms(self.handle, int(on)))

    @property
    def n_emit_launches(self):
        """emission-kernel launches per run (overlap groups; 1 in fused mode)"""
        return int(lib().ed_batch_n_emit_launches(self.handle))

    def fit(self, test, ref, phi_out, expected_out, stream=None, by=1):
        """Per-sample beta-binomial fit (phi, expected) -- counterpart of aod::betabin at reference
        R/class_definition.R:118.  phi_out/expected_out: device float64[n_samples].
        by > 1: fit on exons 0, by, 2*by, ... only (scalar subset.for.speed = n  <=>  by = n_exons // n, :107-113)."""
        keep = []
        cdt = getattr(self, "_cdt", np.int32)
        pt = _device_pointer(test, cdt, keep)
        pr = _device_pointer(ref, cdt, keep)
        pp = _device_pointer(phi_out, np.float64, keep)
        pe = _device_pointer(expected_out, np.float64, keep)
        self._keep_fit = keep
        check(lib().ed_batch_fit_subset(self.handle, pt, pr, int(by), pp, pe, C.c_void_p(stream or 0)))

    def fit_unconverged(self):
        """(number of samples whose last fit() did not converge, first such sample or -1); synchronises the fit"""
        n, first = C.c_int64(0), C.c_int32(-1)
        check(lib().ed_batch_fit_n_unconverged(self.handle, C.byref(n), C.byref(first)))
        return n.value, first.value

    def run(self, test, ref, phi, expected, mixture=1.0, stream=None):
        """Emissions + Viterbi + call table.  Arguments may be torch CUDA tensors (used in place),
        DeviceArrays, or host arrays (uploaded).  Asynchronous on `stream`."""
        keep = []
        cdt = getattr(self, "_cdt", np.int32)
        pt = _device_pointer(test, cdt, keep)
        pr = _device_pointer(ref, cdt, keep)
        pp = _device_pointer(phi, np.float64, keep)
        pe = _device_pointer(expected, np.float64, keep)
        self._keep_run = keep  # keep temporaries alive until the next run
        check(lib().ed_batch_run(self.handle, pt, pr, pp, pe, float(mixture), C.c_void_p(stream or 0)))

    @property
    def fit_bins_form(self):
        """form the last fit_bins took: 1 = count histograms, 0 = per cell (set_fit_histograms(0), or data beyond the bins)"""
        return int(lib().ed_batch_fit_bins_form(self.handle))

    def fit_bins_unconverged(self):
        """samples the last depth-binned fit left short of its tolerance (the binned Newton's own flags)"""
        n = C.c_int64(0)
        check(lib().ed_batch_fit_bins_n_unconverged(self.handle, C.byref(n)))
        return n.value

    def fit_bins(self, test, ref, phi_bins, phi_bins_out, edges_out, expected_out, stream=None):
        """phi.bins > 1 (reference R/class_definition.R:120-147): per sample the depth levels of the reference
        counts (edges_out: (phi_bins + 1, n_samples) complete.bins), one dispersion per level (phi_bins_out:
        (phi_bins, n_samples)) and the common expected proportion.  Raises EdError("Binning did not happen
        properly ...") like the reference's stop() when a level is empty.  Synchronous."""
        keep = []
        pt = _device_pointer(test, np.int32, keep)
        pr = _device_pointer(ref, np.int32, keep)
        pp = _device_pointer(phi_bins_out, np.float64, keep)
        pg = _device_pointer(edges_out, np.float64, keep)
        pe = _device_pointer(expected_out, np.float64, keep)
        self._keep_fit = keep
        check(lib().ed_batch_fit_bins(self.handle, pt, pr, int(phi_bins), pp, pg, pe, C.c_void_p(stream or 0)))

    def run_bins(self, test, ref, phi_bins, phi_bins_dev, edges_dev, expected, mixture=1.0, stream=None):
        """run() with the per-exon dispersion phi.linear of the phi.bins > 1 model (interpolated on the fly)."""
        keep = []
        pt = _device_pointer(test, np.int32, keep)
        pr = _device_pointer(ref, np.int32, keep)
        pp = _device_pointer(phi_bins_dev, np.float64, keep)
        pg = _device_pointer(edges_dev, np.float64, keep)
        pe = _device_pointer(expected, np.float64, keep)
        self._keep_run = keep
        check(lib().ed_batch_run_bins(self.handle, pt, pr, int(phi_bins), pp, pg, pe, float(mixture), C.c_void_p(stream or 0)))

    def phi_linear(self, ref, phi_bins, phi_bins_dev, edges_dev):
        """(n_exons, n_samples) host array: the S4 `phi` slot of the phi.bins > 1 model."""
        keep = []
        pr = _device_pointer(ref, np.int32, keep)
        pp = _device_pointer(phi_bins_dev, np.float64, keep)
        pg = _device_pointer(edges_dev, np.float64, keep)
        out = DeviceArray(np.zeros((self.plan.n_exons, self.n_samples)))
        check(lib().ed_batch_phi_linear(self.handle, pr, int(phi_bins), pp, pg, out.ptr, None))
        check(lib().ed_synchronize(None))
        return out.to_host().reshape(self.plan.n_exons, self.n_samples)

    def _cov_matrix(self, X):
        """(matrix, K) of the covariates: a host array (n_exons, K), a DeviceArray made from one, or a torch CUDA tensor."""
        if hasattr(X, "data_ptr") or hasattr(X, "ptr"):
            return X, int(X.shape[1])
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(self.plan.n_exons, -1)
        return X, int(X.shape[1])

    def fit_cov(self, test, ref, X, beta_out, phi_out, stream=None):
        """Mean model with covariates (`data` + `formula ~ x1 + ...`, reference R/class_definition.R:86-118): X is the
        (n_exons, K) covariate matrix shared by the samples; beta_out (K + 1, n_samples), phi_out (n_samples).  Synchronous."""
        keep = []
        pt = _device_pointer(test, np.int32, keep)
        pr = _device_pointer(ref, np.int32, keep)
        X, K = self._cov_matrix(X)
        px = _device_pointer(X, np.float64, keep) if K > 0 else None
        pb = _device_pointer(beta_out, np.float64, keep)
        pp = _device_pointer(phi_out, np.float64, keep)
        self._keep_fit = keep
        check(lib().ed_batch_fit_cov(self.handle, pt, pr, px, K, pb, pp, C.c_void_p(stream or 0)))

    def run_cov(self, test, ref, X, beta, phi, mixture=1.0, stream=None):
        """run() with expected = plogis(X beta) per exon."""
        keep = []
        pt = _device_pointer(test, np.int32, keep)
        pr = _device_pointer(ref, np.int32, keep)
        X, K = self._cov_matrix(X)
        px = _device_pointer(X, np.float64, keep) if K > 0 else None
        pb = _device_pointer(beta, np.float64, keep)
        pp = _device_pointer(phi, np.float64, keep)
        self._keep_run = keep
        check(lib().ed_batch_run_cov(self.handle, pt, pr, px, K, pb, pp, float(mixture), C.c_void_p(stream or 0)))

    def expected_cov(self, X, beta):
        """(n_exons, n_samples) host array: the S4 `expected` slot, fitted(mod) = plogis(X beta)."""
        keep = []
        X, K = self._cov_matrix(X)
        px = _device_pointer(X, np.float64, keep) if K > 0 else None
        pb = _device_pointer(beta, np.float64, keep)
        out = DeviceArray(np.zeros((self.plan.n_exons, self.n_samples)))
        check(lib().ed_batch_expected_cov(self.handle, px, K, pb, out.ptr, None))
        check(lib().ed_synchronize(None))
        return out.to_host().reshape(self.plan.n_exons, self.n_samples)

    # ---- results ----
    def n_calls(self):
        n = C.c_int64(0)
        check(lib().ed_batch_n_calls(self.handle, C.byref(n)))
        return n.value

    def n_gsl_errors(self):
        n = C.c_int64(0)
        check(lib().ed_batch_n_gsl_errors(self.handle, C.byref(n)))
        return n.value

    def calls(self):
        n = self.n_calls()
        out = np.zeros(n, dtype=CALL_DTYPE)
        check(lib().ed_batch_copy_calls(self.handle, _ptr(out), n))
        return out

    def call_info(self):
        """BF, reads.expected, reads.observed, reads.ratio of every call (R/class_definition.R:379-405)."""
        n = self.n_calls()
        out = np.zeros(n, dtype=CALL_INFO_DTYPE)
        check(lib().ed_batch_copy_call_info(self.handle, _ptr(out), n))
        return out

    def path(self):
        out = np.empty((self.plan.n_exons, self.n_samples), dtype=np.uint8)
        check(lib().ed_batch_copy_path(self.handle, _ptr(out)))
        return out

    def loglik(self):
        """(n_exons, 3, n_samples): [:,0,:] deletion, [:,1,:] normal, [:,2,:] duplication."""
        out = np.empty((self.plan.n_exons, 3, self.n_samples), dtype=np.float64)
        check(lib().ed_batch_copy_loglik(self.handle, _ptr(out)))
        return out

    def verify_emissions(self, test, ref, phi, expected, mixture=1.0, cap=16):
        """Device-side self-check of the last run()'s likelihood matrix: every cell re-evaluated with the reference's own
        per-cell loop (no hoisting, tables or binning) and compared bit for bit.  Arguments: what run() was given.
        Returns (values compared, values that differ, first mismatches as a list of dicts)."""
        keep = []
        pt = _device_pointer(test, np.int32, keep)
        pr = _device_pointer(ref, np.int32, keep)
        pp = _device_pointer(phi, np.float64, keep)
        pe = _device_pointer(expected, np.float64, keep)
        first = (_lib.EdEmitMismatch * max(int(cap), 1))()
        ncmp, nbad = C.c_int64(0), C.c_int64(0)
        check(lib().ed_batch_verify_emissions(self.handle, pt, pr, pp, pe, float(mixture), C.byref(ncmp), C.byref(nbad),
                                              C.cast(first, C.c_void_p), int(cap)))
        rec = [{f: getattr(first[i], f) for f in ("exon", "sample", "state", "observed", "total", "got", "want")}
               for i in range(min(int(cap), nbad.value))]
        return ncmp.value, nbad.value, rec

    def set_emit_mode(self, mode, cap_obs=None, cap_ref=None, reach=None, tails=None):
        """mode 0 / "strict": GSL's arithmetic operation for operation (default); 1 / "tables": log-gamma difference tables per
        (sample, state) -- three gathers and a sum per cell, ~1e-14 relative (see ed_batch_set_emit_mode)."""
        m = {"strict": 0, "tables": 1, "tables-sm": 2}.get(mode, mode)
        if cap_obs is not None or cap_ref is not None or reach is not None:
            check(lib().ed_batch_set_emit_tables(self.handle, int(cap_obs or 4096), int(cap_ref or 32768), float(reach or 8.0)))
        if tails is not None:
            self.set_emit_tails(tails)
        check(lib().ed_batch_set_emit_mode(self.handle, int(m)))

    def set_counts_layout(self, layout):
        """0: device count matrices are [n_exons][n_samples]; 1: [n_samples][n_exons] (R's column-major matrix; fit + emit mode 2)"""
        check(lib().ed_batch_set_counts_layout(self.handle, int(layout)))

    def set_counts_bits(self, bits):
        """32 (default): int32 device counts; 16: uint16, (n_samples, n_exons) -- counts_layout 1 + emit mode 2 only (ed_batch_set_counts_bits)"""
        check(lib().ed_batch_set_counts_bits(self.handle, int(bits)))
        self._cdt = np.uint16 if int(bits) == 16 else np.int32

    def verify_emissions_tol(self, test, ref, phi, expected, mixture=1.0, rel_tol=1e-10, abs_tol=1e-12, cap=16):
        """verify_emissions with a tolerance: returns a dict(compared, beyond, max_rel, max_abs, first)."""
        keep = []
        cdt = getattr(self, "_cdt", np.int32)
        pt = _device_pointer(test, cdt, keep)
        pr = _device_pointer(ref, cdt, keep)
        pp = _device_pointer(phi, np.float64, keep)
        pe = _device_pointer(expected, np.float64, keep)
        first = (_lib.EdEmitMismatch * max(int(cap), 1))()
        ncmp, nbad, mr, ma = C.c_int64(0), C.c_int64(0), C.c_double(0), C.c_double(0)
        check(lib().ed_batch_verify_emissions_tol(self.handle, pt, pr, pp, pe, float(mixture), float(rel_tol), float(abs_tol), C.byref(ncmp),
                                                  C.byref(nbad), C.byref(mr), C.byref(ma), C.cast(first, C.c_void_p), int(cap)))
        rec = [{f: getattr(first[i], f) for f in ("exon", "sample", "state", "observed", "total", "got", "want")}
               for i in range(min(int(cap), nbad.value))]
        return {"compared": ncmp.value, "beyond": nbad.value, "max_rel": mr.value, "max_abs": ma.value, "first": rec}

    def emit_tables(self, sample):
        """emit mode 1: (Ly, Lr, obs table [Ly][3], ref table [Lr][3], tot table [Ly + Lr][3]) of one sample, as the last run built them"""
        dims = (C.c_int32 * 2)()
        check(lib().ed_batch_copy_emit_tables(self.handle, int(sample), dims, None, 0))
        ly, lr = int(dims[0]), int(dims[1])
        ent = np.empty((2 * (ly + lr), 3))
        if ent.size:
            check(lib().ed_batch_copy_emit_tables(self.handle, int(sample), dims, _ptr(ent), ent.shape[0]))
        return ly, lr, ent[:ly], ent[ly:ly + lr], ent[ly + lr:]

    def n_cold_cells(self):
        n = C.c_int64(0)
        check(lib().ed_batch_n_cold_cells(self.handle, C.byref(n)))
        return n.value

    TABLE_STATS = ("n_cold_cells", "n_samples_without_tables", "cold_list_overflow", "n_cells_without_tables")

    def table_stats(self):
        """table-driven modes: what the last run left to the strict arithmetic (ed_batch_table_stats)"""
        v = (C.c_int64 * 4)()
        check(lib().ed_batch_table_stats(self.handle, v))
        return dict(zip(self.TABLE_STATS, (int(x) for x in v)))

    def table_dims(self, sample):
        """(Ly, Lr, Tm1, w) of one sample in the last run (ed_batch_copy_table_dims): w = N0 of the ref = 0 rule, or the reason when Ly = 0"""
        d = (C.c_int32 * 4)()
        check(lib().ed_batch_copy_table_dims(self.handle, int(sample), d))
        return tuple(int(x) for x in d)

    def table_windows(self, sample):
        """(n1, n2, n3, tail) of one sample in the last run (ed_batch_copy_table_windows): the LDS windows of the sample-major table mode and whether
        the sample is a tail sample (Stirling's series beyond the windows)"""
        d = (C.c_int32 * 4)()
        check(lib().ed_batch_copy_table_windows(self.handle, int(sample), d))
        return tuple(int(x) for x in d)

    def set_emit_tails(self, on=True):
        check(lib().ed_batch_set_emit_tails(self.handle, int(bool(on))))

    def device_pointers(self):
        L = lib()
        return {"loglik": L.ed_batch_loglik(self.handle), "path": L.ed_batch_path(self.handle),
                "calls": L.ed_batch_calls(self.handle)}

    def stage_ms_total(self):
        """(dict of stage -> summed milliseconds, timed runs, timed fits) since enable_timing()"""
        ms = (C.c_double * 5)()
        nr, nf = C.c_int64(0), C.c_int64(0)
        check(lib().ed_batch_stage_ms_total(self.handle, ms, C.byref(nr), C.byref(nf)))
        return dict(zip(self.STAGES, [float(v) for v in ms])), nr.value, nf.value

    def stage_ms(self):
        ms = (C.c_float * 5)()
        check(lib().ed_batch_stage_ms(self.handle, ms))
        return dict(zip(self.STAGES, [float(v) for v in ms]))


class PinnedArray:
    """A numpy array in pinned host memory (ed_host_alloc): the DMA engine reads it in place, no staging copy."""

    def __init__(self, shape, dtype):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(int(x) for x in np.atleast_1d(shape))
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = C.c_void_p()
        check(lib().ed_host_alloc(C.byref(self.ptr), self.nbytes))
        buf = (C.c_char * max(self.nbytes, 1)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().ed_host_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _host_slab(a, layout):
    """a host count slab in the memory layout the C entry reads: layout 0 -- rows of consecutive samples, any row pitch (a window of
    the first columns of a wider matrix stays in place); layout 1 -- dense (n, n_exons).  Anything else (Fortran order, strided
    columns, negative strides) is copied into that form rather than uploaded as it lies."""
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("a count slab is a 2-d array")
    if layout == 0:
        ok = (a.shape[1] <= 1 or a.strides[1] == a.itemsize) and a.strides[0] % a.itemsize == 0 and (a.shape[0] <= 1 or a.strides[0] >= a.shape[1] * a.itemsize)
        return a if ok or a.size == 0 else np.ascontiguousarray(a)
    if layout == 1:
        return a if a.flags.c_contiguous else np.ascontiguousarray(a)
    raise ValueError("layout is 0 ((n_exons, n) arrays) or 1 ((n, n_exons) arrays = R's column-major matrix)")


class Cohort:
    """Slabs of a cohort through the library's pipeline (ed_cohort_*): the loop of reference vignette/vignette.Rnw:390-431
    -- new('ExomeDepth') + CallCNVs() per sample -- for slabs of samples, on streams the library owns.

    submit() / submit_host() return a ticket; batch(ticket) gives a Batch view holding that slab's results (valid until
    `slabs_in_flight` further slabs have been submitted).  run_host() does a whole host-resident cohort."""

    STAGES = Batch.STAGES

    def __init__(self, plan, slab_samples, slabs_in_flight=2, **options):
        self.plan = plan
        self.slab_samples = int(slab_samples)
        self.slabs_in_flight = int(slabs_in_flight)
        self.handle = C.c_void_p()
        self._keep = {}
        check(lib().ed_cohort_create(C.byref(self.handle), plan.handle, self.slab_samples, self.slabs_in_flight))
        self.phi_bins = 1
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        check(lib().ed_cohort_set_option(self.handle, name.encode(), float(value)))
        if name == "counts_bits":
            self._counts_bits = int(value)
        if name == "phi_bins":
            self.phi_bins = int(value)

    def close(self):
        if self.handle:
            lib().ed_cohort_destroy(self.handle)
            self.handle = C.c_void_p()
        self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().ed_cohort_stream(self.handle)

    @property
    def n_emit_launches(self):
        return int(lib().ed_cohort_n_emit_launches(self.handle))

    def submit(self, test, ref, phi=None, expected=None, mixture=1.0, ready_stream=None, n_samples=None):
        """one slab, counts on the device: (n_exons, n) int32 torch CUDA tensors / DeviceArrays (host arrays are uploaded
        synchronously first -- use submit_host for the staged path).  phi / expected: device float64[n] or None (fit)."""
        keep = []
        if n_samples is None:
            n_samples = int(test.shape[1])
        cdt = np.uint16 if getattr(self, "_counts_bits", 32) == 16 else np.int32
        pt = _device_pointer(test, cdt, keep)
        pr = _device_pointer(ref, cdt, keep)
        pp = _device_pointer(phi, np.float64, keep) if phi is not None else None
        pe = _device_pointer(expected, np.float64, keep) if expected is not None else None
        t = C.c_int64(-1)
        check(lib().ed_cohort_submit(self.handle, pt, pr, int(n_samples), pp, pe, float(mixture), C.c_void_p(ready_stream or 0), C.byref(t)))
        # the slab's counts are read until its results have been collected (the call decoration): keep them alive that long
        self._keep[t.value % self.slabs_in_flight] = keep + [test, ref, phi, expected]
        return t.value

    def submit_host(self, test, ref, layout, phi=None, expected=None, mixture=1.0, n_samples=None, row_stride=None):
        """one slab from host memory.  layout 0: (n_exons, n) sample-minor arrays (or a window of the first n columns of a wider
        matrix: pass n_samples and row_stride); layout 1: R's column-major n_exons x n matrix, i.e. a C-contiguous (n, n_exons)
        array.  dtype int32 or uint16 (the 16-bit wire format).  numpy arrays or PinnedArray.array views."""
        test, ref = _host_slab(test, layout), _host_slab(ref, layout)
        if test.dtype != ref.dtype or test.dtype not in (np.dtype(np.int32), np.dtype(np.uint16)):
            raise ValueError("test and ref must both be int32 or both uint16")
        wire = test.dtype.itemsize
        if n_samples is None:
            n_samples = int(test.shape[1] if layout == 0 else test.shape[0])
        if row_stride is None:
            if layout == 0 and test.strides[0] != ref.strides[0]:
                raise ValueError("test and ref windows must share their row stride")
            row_stride = int(test.strides[0] // wire) if layout == 0 else 0
        keep = []
        pp = _device_pointer(phi, np.float64, keep) if phi is not None else None
        pe = _device_pointer(expected, np.float64, keep) if expected is not None else None
        t = C.c_int64(-1)
        check(lib().ed_cohort_submit_host(self.handle, C.c_void_p(test.ctypes.data), C.c_void_p(ref.ctypes.data), int(n_samples),
                                          int(layout), int(wire), int(row_stride), pp, pe, float(mixture), C.byref(t)))
        self._keep[t.value % self.slabs_in_flight] = keep + [test, ref]
        return t.value

    def submit_host_test(self, test, ref, layout, phi=None, expected=None, mixture=1.0, n_samples=None, row_stride=None):
        """one slab whose TEST counts come from host memory (as submit_host) and whose references are on the device already
        (ref: int32 CUDA tensor / DeviceArray in the cohort's DEVICE layout -- (n_exons, n) with counts_layout 0, e.g. a window of
        cohort_select_reference_sets' aggregate references; (n, n_exons) with counts_layout 1)"""
        test = _host_slab(test, layout)
        if test.dtype not in (np.dtype(np.int32), np.dtype(np.uint16)):
            raise ValueError("test must be int32 or uint16")
        wire = test.dtype.itemsize
        if n_samples is None:
            n_samples = int(test.shape[1] if layout == 0 else test.shape[0])
        if row_stride is None:
            row_stride = int(test.strides[0] // wire) if layout == 0 else 0
        keep = []
        pr = _device_pointer(ref, np.int32, keep)
        pp = _device_pointer(phi, np.float64, keep) if phi is not None else None
        pe = _device_pointer(expected, np.float64, keep) if expected is not None else None
        t = C.c_int64(-1)
        check(lib().ed_cohort_submit_host_test(self.handle, C.c_void_p(test.ctypes.data), pr, int(n_samples), int(layout), int(wire),
                                               int(row_stride), pp, pe, float(mixture), C.byref(t)))
        self._keep[t.value % self.slabs_in_flight] = keep + [test, ref]
        return t.value

    def batch(self, ticket):
        """(Batch view, device pointer of phi, device pointer of expected) of a ticket"""
        h, pp, pe = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().ed_cohort_batch(self.handle, int(ticket), C.byref(h), C.byref(pp), C.byref(pe)))
        b = Batch._view(self.plan, int(lib().ed_batch_n_samples(h)), h.value)   # the ticket's own width (a short last slab has its own batch)
        return b, pp.value, pe.value

    def results(self, ticket, n_samples, path=False, loglik=False, info=True):
        """host copies of a ticket's results: dict(calls, info, phi, expected[, path][, loglik]); with the option phi_bins = B > 1
        phi_bins (B, n) and edges (B + 1, n) stand where phi is"""
        b, pp, pe = self.batch(ticket)
        b.n_samples = int(n_samples)
        out = {"calls": b.calls()}
        if info:
            out["info"] = b.call_info()
        exp = np.empty(n_samples)
        if self.phi_bins > 1:
            phib, edges = np.empty((self.phi_bins, n_samples)), np.empty((self.phi_bins + 1, n_samples))
            check(lib().ed_cohort_copy_bins_params(self.handle, int(ticket), _ptr(phib), _ptr(edges), _ptr(exp)))
            out["phi_bins"], out["edges"] = phib, edges
        else:
            phi = np.empty(n_samples)
            check(lib().ed_cohort_copy_params(self.handle, int(ticket), _ptr(phi), _ptr(exp)))
            out["phi"] = phi
        out["expected"] = exp
        if path:
            out["path"] = b.path()
        if loglik:
            out["loglik"] = b.loglik()
        return out

    def wait(self, ticket):
        check(lib().ed_cohort_wait(self.handle, int(ticket)))

    def drain(self):
        check(lib().ed_cohort_drain(self.handle))

    def stage_ms_total(self):
        ms = (C.c_double * 5)()
        nr, nf = C.c_int64(0), C.c_int64(0)
        check(lib().ed_cohort_stage_ms_total(self.handle, ms, C.byref(nr), C.byref(nf)))
        return dict(zip(self.STAGES, [float(v) for v in ms])), nr.value, nf.value

    def n_wide_slabs(self):
        """host-fed int32 slabs that held a count outside 0 .. 65 535 and went up 32 bits wide (ed_cohort_n_wide_slabs)"""
        n = C.c_int64(0)
        check(lib().ed_cohort_n_wide_slabs(self.handle, C.byref(n)))
        return n.value

    def emission_intervals(self):
        """option timing: (n, 2) array of (start_ms, end_ms) of the emission stage of every timed run, relative to one reference event
        (ed_cohort_emission_intervals) -- with several lanes the launches overlap; their union is the chip time they take"""
        n = C.c_int64(0)
        check(lib().ed_cohort_emission_intervals(self.handle, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 0), 2), dtype=np.float32)
        if n.value > 0:
            check(lib().ed_cohort_emission_intervals(self.handle, _ptr(out), n.value, C.byref(n)))
        return out

    def ingest_stats(self):
        b, s = C.c_double(0), C.c_double(0)
        check(lib().ed_cohort_ingest_stats(self.handle, C.byref(b), C.byref(s)))
        return b.value, s.value

    def run_host(self, test, ref, layout, phi=None, expected=None, mixture=1.0, want_path=False):
        """CallCNVs for a whole host-resident cohort (ed_cohort_run_host).  layout 0: (n_exons, S) arrays; layout 1: (S, n_exons)
        arrays (R's column-major n_exons x S matrix).  Returns dict(calls, info, phi, expected, n_unconverged, n_gsl_errors[, path])."""
        test, ref = np.ascontiguousarray(test), np.ascontiguousarray(ref)
        if test.dtype != ref.dtype or test.dtype not in (np.dtype(np.int32), np.dtype(np.uint16)):
            raise ValueError("test and ref must both be int32 or both uint16")
        S = int(test.shape[1] if layout == 0 else test.shape[0])
        E = self.plan.n_exons
        ph = _f64(phi) if phi is not None else None
        ex = _f64(expected) if expected is not None else None
        phi_out, exp_out = np.empty(S), np.empty(S)
        path = np.empty((E, S) if layout == 0 else (S, E), dtype=np.uint8) if want_path else None
        n = C.c_int64(0)
        check(lib().ed_cohort_run_host(self.handle, C.c_void_p(test.ctypes.data), C.c_void_p(ref.ctypes.data), S, int(layout),
                                       int(test.dtype.itemsize), _ptr(ph) if ph is not None else None, _ptr(ex) if ex is not None else None,
                                       float(mixture), _ptr(phi_out), _ptr(exp_out), _ptr(path) if path is not None else None, C.byref(n)))
        calls = np.zeros(n.value, dtype=CALL_DTYPE)
        info = np.zeros(n.value, dtype=CALL_INFO_DTYPE)
        check(lib().ed_cohort_copy_calls(self.handle, _ptr(calls), _ptr(info), n.value))
        nu, ne = C.c_int64(0), C.c_int64(0)
        check(lib().ed_cohort_run_status(self.handle, C.byref(nu), C.byref(ne)))
        out = {"calls": calls, "info": info, "phi": phi_out, "expected": exp_out, "n_unconverged": nu.value, "n_gsl_errors": ne.value}
        ts = (C.c_int64 * 4)()
        check(lib().ed_cohort_table_status(self.handle, ts))
        out["table_stats"] = dict(zip(Batch.TABLE_STATS, (int(x) for x in ts)))
        if self.phi_bins > 1:
            del out["phi"]
            out["phi_bins"], out["edges"] = np.empty((self.phi_bins, S)), np.empty((self.phi_bins + 1, S))
            check(lib().ed_cohort_copy_bins(self.handle, _ptr(out["phi_bins"]), _ptr(out["edges"])))
        if want_path:
            out["path"] = path
        return out


class MultiDevice:
    """ed_cohort_run_host over several devices from ONE process (ed_multi_*; csrc/edmulti.inc): the exon design replicated per
    device, the slabs dealt from one queue, one host thread per device, call tables put together in column order.
    devices=None: every visible device once; a device may be listed more than once."""

    def __init__(self, chrom_off, start, end, slab_samples, devices=None, slabs_in_flight=2, transition_probability=1e-4,
                 expected_cnv_length=50000.0, **options):
        chrom_off = np.ascontiguousarray(chrom_off, dtype=np.int32)
        start = np.ascontiguousarray(start, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        self.n_exons = int(start.size)
        self.handle = C.c_void_p()
        dv = np.ascontiguousarray(devices, dtype=np.int32) if devices is not None else None
        check(lib().ed_multi_create(C.byref(self.handle), _ptr(dv) if dv is not None else None, int(dv.size) if dv is not None else 0,
                                    self.n_exons, int(chrom_off.size - 1), _ptr(chrom_off), _ptr(start), _ptr(end),
                                    float(transition_probability), float(expected_cnv_length), int(slab_samples), int(slabs_in_flight)))
        self.phi_bins = 1
        for k, v in options.items():
            self.set_option(k, v)

    @property
    def n_devices(self):
        return int(lib().ed_multi_n_devices(self.handle))

    def set_option(self, name, value):
        check(lib().ed_multi_set_option(self.handle, name.encode(), float(value)))
        if name == "phi_bins":
            self.phi_bins = int(value)

    def close(self):
        if self.handle:
            lib().ed_multi_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_host(self, test, ref, layout, phi=None, expected=None, mixture=1.0, want_path=False):
        """as Cohort.run_host; the result also carries `devices`: per device the slabs / columns it took from the queue, its thread's seconds, its NUMA node"""
        test, ref = np.ascontiguousarray(test), np.ascontiguousarray(ref)
        if test.dtype != ref.dtype or test.dtype not in (np.dtype(np.int32), np.dtype(np.uint16)):
            raise ValueError("test and ref must both be int32 or both uint16")
        S = int(test.shape[1] if layout == 0 else test.shape[0])
        E = self.n_exons
        ph = _f64(phi) if phi is not None else None
        ex = _f64(expected) if expected is not None else None
        phi_out, exp_out = np.empty(S), np.empty(S)
        path = np.empty((E, S) if layout == 0 else (S, E), dtype=np.uint8) if want_path else None
        n = C.c_int64(0)
        check(lib().ed_multi_run_host(self.handle, C.c_void_p(test.ctypes.data), C.c_void_p(ref.ctypes.data), S, int(layout),
                                      int(test.dtype.itemsize), _ptr(ph) if ph is not None else None, _ptr(ex) if ex is not None else None,
                                      float(mixture), _ptr(phi_out), _ptr(exp_out), _ptr(path) if path is not None else None, C.byref(n)))
        calls = np.zeros(n.value, dtype=CALL_DTYPE)
        info = np.zeros(n.value, dtype=CALL_INFO_DTYPE)
        check(lib().ed_multi_copy_calls(self.handle, _ptr(calls), _ptr(info), n.value))
        nu, ne = C.c_int64(0), C.c_int64(0)
        check(lib().ed_multi_run_status(self.handle, C.byref(nu), C.byref(ne)))
        out = {"calls": calls, "info": info, "phi": phi_out, "expected": exp_out, "n_unconverged": nu.value, "n_gsl_errors": ne.value}
        ts = (C.c_int64 * 4)()
        check(lib().ed_multi_table_status(self.handle, ts))
        out["table_stats"] = dict(zip(Batch.TABLE_STATS, (int(x) for x in ts)))
        if self.phi_bins > 1:
            del out["phi"]
            out["phi_bins"], out["edges"] = np.empty((self.phi_bins, S)), np.empty((self.phi_bins + 1, S))
            check(lib().ed_multi_copy_bins(self.handle, _ptr(out["phi_bins"]), _ptr(out["edges"])))
        if want_path:
            out["path"] = path
        D = self.n_devices
        dv, ns, nc, sec, nn = (C.c_int * D)(), (C.c_int64 * D)(), (C.c_int64 * D)(), (C.c_double * D)(), (C.c_int * D)()
        check(lib().ed_multi_device_stats(self.handle, dv, ns, nc, sec, nn))
        out["devices"] = [{"device": int(dv[i]), "slabs": int(ns[i]), "columns": int(nc[i]), "seconds": float(sec[i]), "numa_node": int(nn[i])} for i in range(D)]
        return out


# ---------------------------------------------------------------------------------------------
# exon ordering of CallCNVs (reference R/class_definition.R:323-336)
# ---------------------------------------------------------------------------------------------
def chromosome_order(chromosome, start, end):
    """Return (order, chrom_levels, chrom_codes_sorted, chrom_off).  Levels are '1'..'22' first (those
    present), then the other names in first-seen order; exons are ordered by (level, midpoint) with
    ties kept in input order, as R's order() does."""
    chromosome = np.asarray([str(c) for c in chromosome], dtype=object)
    used = []
    seen = set()
    for c in chromosome:
        if c not in seen:
            seen.add(c)
            used.append(c)
    autos = [str(i) for i in range(1, 23)]
    levels = [c for c in autos if c in seen] + [c for c in used if c not in autos]
    code_of = {c: i for i, c in enumerate(levels)}
    codes = np.fromiter((code_of[c] for c in chromosome), dtype=np.int64, count=chromosome.size)
    mid = 0.5 * (np.asarray(start, dtype=np.float64) + np.asarray(end, dtype=np.float64))
    order = np.lexsort((mid, codes))  # stable: last key is primary
    sc = codes[order]
    chrom_off = np.zeros(len(levels) + 1, dtype=np.int32)
    for i in range(len(levels)):
        chrom_off[i + 1] = chrom_off[i] + int(np.sum(sc == i))
    return order, levels, sc, chrom_off


class ExomeDepth:
    """Mirror of the reference's S4 class (R/class_definition.R:36-46) for one test sample.

    phi / expected: if given, the fixed dispersion and expected proportion (the reference gets them
    from aod::betabin, :118, :168); if omitted they are fitted on the GPU (ed_batch_fit)."""

    def __init__(self, test, reference, phi=None, expected=None, prop_tumor=1.0, subset_for_speed=None, phi_bins=1,
                 data=None, formula="cbind(test, reference) ~ 1", verbose=False):
        test = np.asarray(test, dtype=np.float64)
        reference = np.asarray(reference, dtype=np.float64)
        if test.size != reference.size:
            raise ValueError("Length of test and numeric must match")   # R/class_definition.R:92
        self.test, self.reference = test, reference
        self.prop_tumor = float(prop_tumor)
        self.phi = np.zeros(0)
        self.expected = np.zeros(0)
        self.likelihood = np.zeros((0, 3))
        self.annotations = None
        self.CNV_calls = None
        self.cor_test_reference = None
        if np.sum(test > 5) < 5:                                       # R/class_definition.R:94-97
            if verbose:
                print("It looks like the test samples has only %d bins with more than 5 reads." % np.sum(test > 5))
            return
        n = test.size
        self.formula = formula
        terms = _formula_terms(formula)
        if (phi is None or expected is None) and terms:                 # covariates: `data` columns named in the formula
            if phi_bins != 1 or subset_for_speed is not None:
                raise NotImplementedError("covariates together with phi.bins > 1 or subset.for.speed are not implemented")
            if data is None:
                raise ValueError("the formula refers to covariates but no `data` was given")
            X = np.stack([np.asarray(data[t], dtype=np.float64) for t in terms], axis=1)
            if X.shape[0] != n:
                raise ValueError("`data` must have one row per exon")
            phi, expected = fit_betabin_cov(_as_r_integer(test), _as_r_integer(reference), X)
        if (phi is None or expected is None) and phi_bins != 1:        # R/class_definition.R:120-147
            if subset_for_speed is not None:
                raise ValueError("Subset for speed option is not compatible with variable phi. This will be fixed later on "
                                 "but for now please adapt your code.")
            phi, expected = fit_betabin_bins(_as_r_integer(test), _as_r_integer(reference), int(phi_bins))
        if phi is None or expected is None:
            rows = np.arange(n)
            if subset_for_speed is not None:                           # R/class_definition.R:107-113
                sub = np.atleast_1d(np.asarray(subset_for_speed))
                if sub.size == 1 and np.issubdtype(sub.dtype, np.number):
                    by = int(np.floor(n / float(sub[0])))
                    if by < 1:
                        raise ValueError("wrong sign in 'by' argument")   # R's seq() error for by = 0
                    rows = np.arange(0, n, by)                         # seq(from = 1, to = nrow, by = floor(nrow / n))
                else:
                    sub = sub.astype(np.int64)
                    rows = sub[(sub >= 1) & (sub <= n)] - 1            # keep existing (1-based) rows only
            phi, expected = fit_betabin(_as_r_integer(test[rows]), _as_r_integer(reference[rows]))
        self.phi = np.full(n, float(phi)) if np.ndim(phi) == 0 else _f64(phi)
        self.expected = np.full(n, float(expected)) if np.ndim(expected) == 0 else _f64(expected)
        self.likelihood = np.array(get_loglike_matrix(self.phi, self.expected, _as_r_integer(reference + test),
                                                      _as_r_integer(test), mixture=prop_tumor))

    def TestCNV(self, chromosome, start, end, type):
        """reference R/class_definition.R:243-256 (needs CallCNVs annotations for the positions)."""
        if type not in ("deletion", "duplication"):
            raise ValueError("type must be either duplication or deletion\n")
        if self.annotations is None:
            raise ValueError("This function cannot be used if the position of the exons/DNA segments was not included")
        a = self.annotations
        which = (a["chromosome"] == str(chromosome)) & (a["start"] >= start) & (a["end"] <= end)
        col = 0 if type == "deletion" else 2
        return float(np.sum(self.likelihood[which, col] - self.likelihood[which, 1]))

    def CallCNVs(self, chromosome, start, end, name, transition_probability=1e-4, expected_CNV_length=50000):
        """reference R/class_definition.R:311-419: order exons, one Viterbi chain per chromosome,
        call table with start.p/end.p (1-based, global), type, nexons, start, end, chromosome, id.

        One intentional difference on UNSORTED input: the reference reorders x@test, x@reference, x@annotations and
        x@likelihood (:327-336) but not x@expected / x@phi, and then indexes x@expected with the reordered positions
        (:396) -- with a per-exon `expected` (covariates in the formula) its reads.expected is then summed over the wrong
        exons.  Here phi and expected are reordered with everything else, so reads.expected belongs to the call's exons.
        With the default formula (expected constant) the two agree."""
        if self.phi.size == 0:
            self.CNV_calls = []
            return self
        n = self.likelihood.shape[0]
        if not (len(start) == len(chromosome) == len(end) == len(name)):
            raise ValueError("Chromosome, name, start and end vector must have the same lengths.\n")
        if n != len(chromosome):
            raise ValueError("The annotation vectors must have the same length as the data in the ExomeDepth x")
        order, levels, codes, chrom_off = chromosome_order(chromosome, start, end)
        start = np.asarray(start)[order]
        end = np.asarray(end)[order]
        name = np.asarray(name, dtype=object)[order]
        chrom_sorted = np.asarray([str(c) for c in chromosome], dtype=object)[order]
        if np.any(order != np.arange(n)):
            self.test = self.test[order]
            self.reference = self.reference[order]
            self.likelihood = self.likelihood[order]
            self.phi = self.phi[order]
            self.expected = self.expected[order]
        self.annotations = {"name": name, "chromosome": chrom_sorted, "start": start, "end": end}
        self.cor_test_reference = float(np.corrcoef(self.test, self.reference)[0, 1])
        constant = bool(np.all(self.phi == self.phi[0]) and np.all(self.expected == self.expected[0]))
        if constant:
            plan = Plan(chrom_off, start, end, transition_probability, expected_CNV_length)
            batch = Batch(plan, 1)
            try:
                # the likelihood is recomputed on the device from the same inputs -- prop.tumor included -- hence
                # bit-identical to the slot the reference runs its Viterbi on (R/class_definition.R:364)
                batch.run(_as_r_integer(self.test).reshape(n, 1), _as_r_integer(self.reference).reshape(n, 1),
                          self.phi[:1], self.expected[:1], mixture=self.prop_tumor)
                raw = batch.calls()
                self.Viterbi_path = batch.path()[:, 0].astype(np.int64)
            finally:
                batch.close()
                plan.close()
        else:
            # per-exon phi (phi.bins > 1) or expected: the likelihood slot itself goes through the .Call-shaped
            # entry, chromosome by chromosome, exactly as R/class_definition.R:343-374 does
            tp = transition_probability
            T = np.array([[1. - tp, tp / 2., tp / 2.], [0.5, 0.5, 0.], [0.5, 0., 0.5]])
            raw = []
            self.Viterbi_path = np.zeros(n, dtype=np.int64)
            for c in range(len(chrom_off) - 1):
                lo, hi = int(chrom_off[c]), int(chrom_off[c + 1])
                if hi <= lo:
                    continue
                ll = np.vstack([[-np.inf, 0., -np.inf], self.likelihood[lo:hi][:, [1, 0, 2]], [-100., 0., -100.]])
                pos = np.concatenate([[start[lo] - 2 * expected_CNV_length], start[lo:hi],
                                      [end[hi - 1] + 2 * expected_CNV_length]])
                res = viterbi_hmm(T, ll, _as_r_integer(pos.astype(np.float64)), expected_CNV_length)
                self.Viterbi_path[lo:hi] = res["Viterbi.path"][1:-1]
                for r in res["calls"]:
                    raw.append({"start_exon": int(r["start.p"]) - 2 + lo, "end_exon": int(r["end.p"]) - 2 + lo,
                                "type": int(r["type"]), "nexons": int(r["nexons"])})
        calls = []
        total = self.test + self.reference
        for r in raw:
            s, e = int(r["start_exon"]), int(r["end_exon"])
            typ = ["deletion", "duplication"][int(r["type"]) - 1]
            col = 0 if typ == "deletion" else 2
            bf = float(np.sum(self.likelihood[s:e + 1, col] - self.likelihood[s:e + 1, 1]))
            reads_expected = int(np.sum(total[s:e + 1] * self.expected[s:e + 1]))
            reads_observed = float(np.sum(self.test[s:e + 1]))
            with np.errstate(divide="ignore", invalid="ignore"):   # obs / 0 is Inf in R (and in k_call_info), 0 / 0 NaN
                ratio = float(np.float64(reads_observed) / np.float64(reads_expected))
            cid = ("chr%s:%d-%d" % (chrom_sorted[s], start[s], end[e])).replace("chrchr", "chr")
            calls.append({"start.p": s + 1, "end.p": e + 1, "type": typ, "nexons": int(r["nexons"]),
                          "start": int(start[s]), "end": int(end[e]), "chromosome": chrom_sorted[s], "id": cid,
                          "BF": _signif(np.log10(np.e) * bf, 3), "reads.expected": reads_expected,
                          "reads.observed": reads_observed,
                          "reads.ratio": _signif(ratio, 3)})
        self.CNV_calls = calls
        return self


REFSET_DTYPE = np.dtype([("ref_index", "<i4"), ("selected", "<i4"), ("correlation", "<f8"), ("expected_BF", "<f8"),
                         ("phi", "<f8"), ("ratio_sd", "<f8"), ("mean_p", "<f8"), ("median_depth", "<f8")])
assert REFSET_DTYPE.itemsize == C.sizeof(EdRefsetRow)


def select_reference_set(test_counts, reference_counts, bin_length=None, n_bins_reduced=0, names=None, prefix_window=None):
    """reference R/optimize_reference_set.R:53-148 (formula ~ 1, phi.bins = 1).

    test_counts: (E,) ; reference_counts: (E, R) matrix, one column per candidate reference sample (host array,
    or a torch CUDA int32 tensor of shape (E, R)).  Returns {'reference.choice': [...], 'summary.stats':
    structured array (REFSET_DTYPE, sorted by decreasing correlation), 'n.bins': int}.
    prefix_window=(begin, end): raw statistics of those cumulative references only (a rank's share, see
    dist.select_reference_set_sharded); 'reference.choice' is then None until refset_finalize()."""
    keep = []
    if hasattr(reference_counts, "shape") and len(reference_counts.shape) != 2:
        raise ValueError("The reference sequence count data must be provided as a matrix")
    E, R = int(reference_counts.shape[0]), int(reference_counts.shape[1])
    if int(np.prod(np.shape(test_counts))) != E and not hasattr(test_counts, "data_ptr"):
        raise ValueError("The number of rows of the reference matrix must match the length of the test count data\n")
    if not hasattr(test_counts, "data_ptr"):
        test_counts = _as_r_integer(np.asarray(test_counts))
    if not hasattr(reference_counts, "data_ptr"):
        reference_counts = _as_r_integer(np.asarray(reference_counts))
    pt = _device_pointer(test_counts, np.int32, keep)
    pr = _device_pointer(reference_counts, np.int32, keep)
    bl = None
    if bin_length is not None:
        bl = _f64(bin_length)
        if np.any(bl == 0):
            z = int(np.sum(bl == 0))
            raise ValueError("bin.length contains %d zero%s. This causes NAs in correlation computing. All bin lengths "
                             "must be positive" % (z, "s" if z > 1 else ""))
    rows = np.zeros(R, dtype=REFSET_DTYPE)
    n_chosen = C.c_int32(0)
    n_sel = C.c_int64(0)
    if prefix_window is not None:
        check(lib().ed_select_reference_set_part(pt, pr, E, R, _ptr(bl) if bl is not None else None, int(n_bins_reduced),
                                                 int(prefix_window[0]), int(prefix_window[1]), _ptr(rows),
                                                 C.byref(n_chosen), C.byref(n_sel), None))
        return {"reference.choice": None, "summary.stats": rows, "n.bins": int(n_sel.value),
                "low.coverage": n_chosen.value == 1}
    check(lib().ed_select_reference_set(pt, pr, E, R, _ptr(bl) if bl is not None else None, int(n_bins_reduced),
                                        _ptr(rows), C.byref(n_chosen), C.byref(n_sel), None))
    if names is None:
        names = ["X%d" % (i + 1) for i in range(R)]       # R/optimize_reference_set.R:76
    choice = [names[int(i)] for i in rows["ref_index"][: n_chosen.value]]
    return {"reference.choice": choice, "summary.stats": rows, "n.bins": int(n_sel.value)}


def refcohort_last_path():
    """ed_refcohort_last_path: how the last cohort_select_reference_sets call formed the statistics of its cumulative references --
    dict(chunks_by_columns, chunks_row_major, columns_beyond_bins, max_newton_iterations, columns_large_geometry)."""
    out = (C.c_int64 * 5)()
    check(lib().ed_refcohort_last_path(out))
    return dict(zip(("chunks_by_columns", "chunks_row_major", "columns_beyond_bins", "max_newton_iterations", "columns_large_geometry"), (int(v) for v in out)))


def cohort_select_reference_sets(counts, bin_length=None, n_bins_reduced=0, max_refs=32, want_reference=True, want_correlations=False,
                                 reference_out=None, test_range=None, sample_major=False, counts_sm_out=None, stream=None, grow_max_refs=True):
    """select.reference.set for every sample of a cohort against all the others (reference vignette/vignette.Rnw:390-402 loop;
    R/optimize_reference_set.R:53-148 per sample), in one call.

    counts: (E, S) int32 host array or torch CUDA tensor.  Returns dict(n_chosen (S,), choice (S, K) -1 padded,
    summary.stats (S, K) REFSET_DTYPE, n.bins[, reference: DeviceArray (E, S) aggregate reference][, correlations (S, S)]).
    test_range = (t0, t1): only the samples t0 <= t < t1 as tests (all S still candidates) -- one rank's share of a sample-sharded
    cohort; the per-test outputs then have t1 - t0 rows / columns (ed_cohort_select_reference_sets_range).
    sample_major=True: the aggregate references come back sample-major, (n_tests, E) -- R's column-major matrix, what Cohort(emit_mode=2,
    counts_layout=1) takes -- and counts_sm_out (an (S, E) int32 device array), when given, receives the count matrix transposed alongside
    (ed_cohort_select_reference_sets_sm).
    stream: the HIP stream (its handle as an integer, e.g. torch.cuda.Stream().cuda_stream) the entry's kernels and copies are issued on; None =
    the null stream, which is ordered against every blocking stream of the process -- a caller that wants a copy on another stream to run
    beside this call names a stream of its own here.  The entry returns when its work is complete either way.
    grow_max_refs: a test whose choice is LONGER than max_refs is an error of the C entry ("call again with a larger max_refs": its outputs
    have max_refs columns); True = this wrapper does that, doubling max_refs until every choice fits (the result never depends on
    max_refs otherwise).  With thousands of candidates (a rank of a sample-sharded cohort: 8 191 of them) choices of 33 - 35 do occur."""
    if grow_max_refs:
        k = int(max_refs if max_refs > 0 else 32)
        while True:
            try:
                return cohort_select_reference_sets(counts, bin_length, n_bins_reduced, k, want_reference, want_correlations, reference_out, test_range,
                                                    sample_major, counts_sm_out, stream, grow_max_refs=False)
            except EdError as e:
                if "larger max_refs" not in str(e) or k >= int(counts.shape[1]) - 1:
                    raise
                k = min(2 * k, int(counts.shape[1]) - 1)
    keep = []
    E, S = int(counts.shape[0]), int(counts.shape[1])
    t0, t1 = (0, S) if test_range is None else (int(test_range[0]), int(test_range[1]))
    Sr = t1 - t0
    if not hasattr(counts, "data_ptr"):
        counts = _as_r_integer(np.asarray(counts))
    pc = _device_pointer(counts, np.int32, keep)
    K = int(min(max_refs if max_refs > 0 else 32, S - 1))
    bl = _f64(bin_length) if bin_length is not None else None
    n_chosen = np.zeros(Sr, dtype=np.int32)
    choice = np.full((Sr, K), -1, dtype=np.int32)
    rows = np.zeros((Sr, K), dtype=REFSET_DTYPE)
    corr = np.zeros((Sr, S)) if want_correlations else None
    ref = DeviceArray(nbytes=E * Sr * 4) if (want_reference and reference_out is None) else None
    if ref is not None:
        ref.host_dtype, ref.shape = np.dtype(np.int32), ((Sr, E) if sample_major else (E, Sr))
    ref_ptr = ref.ptr if ref is not None else None
    if reference_out is not None:          # the caller's own (E, S) int32 device array (a torch CUDA tensor, say) receives the aggregate references
        ref_ptr = _device_pointer(reference_out, np.int32, keep)
    nsel = C.c_int64(0)
    if sample_major:
        if ref_ptr is None:
            raise ValueError("sample_major=True returns the aggregate references: want_reference or reference_out")
        cs_ptr = _device_pointer(counts_sm_out, np.int32, keep) if counts_sm_out is not None else None
        check(lib().ed_cohort_select_reference_sets_sm(pc, E, S, _ptr(bl) if bl is not None else None, int(n_bins_reduced), K, t0, t1,
                                                       _ptr(n_chosen), _ptr(choice), _ptr(rows), _ptr(corr) if corr is not None else None,
                                                       ref_ptr, cs_ptr, C.byref(nsel), C.c_void_p(stream or 0)))
    else:
        check(lib().ed_cohort_select_reference_sets_range(pc, E, S, _ptr(bl) if bl is not None else None, int(n_bins_reduced), K, t0, t1,
                                                          _ptr(n_chosen), _ptr(choice), _ptr(rows), _ptr(corr) if corr is not None else None,
                                                          ref_ptr, C.byref(nsel), C.c_void_p(stream or 0)))
    out = {"n_chosen": n_chosen, "choice": choice, "summary.stats": rows, "n.bins": int(nsel.value)}
    if ref is not None:
        out["reference"] = ref
    elif reference_out is not None:
        out["reference"] = reference_out
    if corr is not None:
        out["correlations"] = corr
    return out


def get_power_betabinom(size, my_phi, my_p, my_alt_p, theory=False, frequentist=False, limit=False):
    """reference R/tools.R:128-166 (vectorised over its arguments): the expected log10 Bayes factor.  theory=True is the
    reference's binomial case (:137-142).  `frequentist` is accepted and ignored, as in the reference (it is never read).
    limit=True (:145-153) is, in the reference, a Monte-Carlo average over 2000 draws of R's generator; here its expectation
    (sum over 0 < x < size of dbetabinom.ab(x; alt) times the log10 ratio of the beta densities at x / size): the same quantity
    without the sampling noise."""
    mode = 1 if theory else (2 if limit else 0)     # (theory wins, as the reference's two `if` blocks have it)
    size, my_phi, my_p, my_alt_p = np.broadcast_arrays(_f64(np.atleast_1d(size)), _f64(np.atleast_1d(my_phi)),
                                                       _f64(np.atleast_1d(my_p)), _f64(np.atleast_1d(my_alt_p)))
    size, my_phi, my_p, my_alt_p = (_f64(a) for a in (size, my_phi, my_p, my_alt_p))
    out = np.empty(size.size, dtype=np.float64)
    check(lib().ed_get_power_betabinom_mode(size.size, _ptr(size), _ptr(my_phi), _ptr(my_p), _ptr(my_alt_p), mode, _ptr(out)))
    return out if out.size > 1 else float(out[0])


def refset_finalize(rows, names=None):
    """Early exit + reference.choice (R/optimize_reference_set.R:130, :143-145) on a complete table of raw rows."""
    rows = np.ascontiguousarray(rows, dtype=REFSET_DTYPE).copy()
    n_chosen = C.c_int32(0)
    check(lib().ed_refset_finalize(_ptr(rows), rows.size, C.byref(n_chosen)))
    if names is None:
        names = ["X%d" % (i + 1) for i in range(rows.size)]
    return {"reference.choice": [names[int(i)] for i in rows["ref_index"][: n_chosen.value]], "summary.stats": rows}


def _signif(x, digits):
    """R's signif() for the call table's BF and reads.ratio (R/class_definition.R:403-404)."""
    if x == 0 or not np.isfinite(x):
        return x
    from math import floor, log10
    e = digits - 1 - int(floor(log10(abs(x))))
    return round(x * 10 ** e) / 10 ** e if e >= 0 else round(x / 10 ** (-e)) * 10 ** (-e)


def fit_betabin_bins(test, reference, phi_bins):
    """phi.bins > 1 for one sample on the GPU: returns (phi.linear per exon, expected)."""
    test = _i32(test)
    reference = _i32(reference)
    n = test.size
    plan = Plan(np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32) + 1)
    batch = Batch(plan, 1)
    try:
        phib = DeviceArray(np.zeros(phi_bins))
        edges = DeviceArray(np.zeros(phi_bins + 1))
        exp = DeviceArray(np.zeros(1))
        t = DeviceArray(test.reshape(n, 1))
        r = DeviceArray(reference.reshape(n, 1))
        batch.fit_bins(t, r, phi_bins, phib, edges, exp)
        phi_lin = batch.phi_linear(r, phi_bins, phib, edges)[:, 0]
        return phi_lin, float(exp.to_host()[0])
    finally:
        batch.close()
        plan.close()


def _formula_terms(formula):
    """Right-hand-side terms of 'cbind(test, reference) ~ x1 + x2' ('1' = intercept only -> [])."""
    if "~" not in formula:
        raise ValueError("formula must look like 'cbind(test, reference) ~ ...'")
    rhs = formula.split("~", 1)[1]
    terms = [t.strip() for t in rhs.split("+")]
    terms = [t for t in terms if t not in ("", "1")]
    for t in terms:
        if not t.replace("_", "").replace(".", "").isalnum():
            raise NotImplementedError("only main effects of numeric covariates are implemented (term %r)" % t)
    return terms


def fit_betabin_cov(test, reference, X):
    """Covariate model for one sample on the GPU: returns (phi, expected per exon)."""
    test = _i32(test)
    reference = _i32(reference)
    n = test.size
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(n, -1)
    plan = Plan(np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32) + 1)
    batch = Batch(plan, 1)
    try:
        beta = DeviceArray(np.zeros((X.shape[1] + 1, 1)))
        phi = DeviceArray(np.zeros(1))
        t = DeviceArray(test.reshape(n, 1))
        r = DeviceArray(reference.reshape(n, 1))
        batch.fit_cov(t, r, X, beta, phi)
        expected = batch.expected_cov(X, beta)[:, 0]
        return float(phi.to_host()[0]), expected
    finally:
        batch.close()
        plan.close()


def fit_betabin(test, reference):
    """Fit (phi, expected) of  cbind(test, reference) ~ 1  for one sample on the GPU."""
    test = _i32(test)
    reference = _i32(reference)
    n = test.size
    plan = Plan(np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32) + 1)
    batch = Batch(plan, 1)
    try:
        phi = DeviceArray(np.zeros(1))
        exp = DeviceArray(np.zeros(1))
        batch.fit(test.reshape(n, 1), reference.reshape(n, 1), phi, exp)
        check(lib().ed_synchronize(None))
        if batch.fit_unconverged()[0]:
            import warnings
            warnings.warn("beta-binomial fit: the Newton iteration did not converge (phi, expected are its last iterate)")
        return float(phi.to_host()[0]), float(exp.to_host()[0])
    finally:
        batch.close()
        plan.close()
