"""ctypes binding of libedcore.so (the C-ABI declared in include/exomedepth_amd.h).

There is deliberately no fallback: if the shared library is missing, or no gfx950 device is usable,
every compute entry raises.  The CPU checker under oracle/ is test infrastructure and is never
imported from here.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libedcore.so")
if os.environ.get("ED_LIB_VARIANT"):      # a diagnostic build of the same sources (_build.VARIANTS: sanitizers, experiments); never the default
    LIB_PATH = os.path.join(HERE, "libedcore_%s.so" % os.environ["ED_LIB_VARIANT"])

ED_OK = 0


class EdError(RuntimeError):
    pass


class EdRefsetRow(C.Structure):
    _fields_ = [("ref_index", C.c_int32), ("selected", C.c_int32), ("correlation", C.c_double), ("expected_BF", C.c_double),
                ("phi", C.c_double), ("ratio_sd", C.c_double), ("mean_p", C.c_double), ("median_depth", C.c_double)]


class EdCallInfo(C.Structure):
    _fields_ = [("BF_raw", C.c_double), ("BF", C.c_double), ("reads_expected", C.c_int64), ("reads_observed", C.c_int64),
                ("reads_ratio", C.c_double)]


class EdEmitMismatch(C.Structure):
    _fields_ = [("exon", C.c_int64), ("sample", C.c_int64), ("state", C.c_int32), ("observed", C.c_int32), ("total", C.c_int32),
                ("pad_", C.c_int32), ("got", C.c_double), ("want", C.c_double)]


class EdCall(C.Structure):
    _fields_ = [("sample", C.c_int32), ("chrom", C.c_int32), ("start_exon", C.c_int32), ("end_exon", C.c_int32),
                ("type", C.c_int32), ("nexons", C.c_int32)]


_lib = None

# every symbol include/exomedepth_amd.h declares: (name, restype, argtypes)
_vp, _i64, _i32, _dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
SYMBOLS = [
    ("ed_version", C.c_char_p, []),
    ("ed_last_error", C.c_char_p, []),
    ("ed_device_count", C.c_int, []),
    ("ed_device_info", C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    ("ed_get_loglike_matrix", C.c_int, [_vp, _vp, _vp, _vp, _i64, _dbl, _vp, C.POINTER(_i64)]),
    ("ed_get_loglike_matrix_messages", C.c_int, [_vp, _vp, _vp, _vp, _i64, _dbl, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ed_hmm", C.c_int, [_i32, _i32, _vp, _vp, _vp, _dbl, _vp, _vp, _i64, C.POINTER(_i64)]),
    ("ed_dropin_release", None, []),
    ("ed_plan_create", C.c_int, [C.POINTER(_vp), C.c_int, _i64, _i32, _vp, _vp, _vp, _dbl, _dbl]),
    ("ed_plan_destroy", None, [_vp]),
    ("ed_plan_n_exons", _i64, [_vp]),
    ("ed_batch_create", C.c_int, [C.POINTER(_vp), _vp, _i64]),
    ("ed_batch_destroy", None, [_vp]),
    ("ed_batch_fit", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    ("ed_batch_set_fit_histograms", C.c_int, [_vp, C.c_int]),
    ("ed_batch_fit_n_unconverged", C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i32)]),
    ("ed_batch_fit_subset", C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, _vp]),
    ("ed_batch_fit_bins", C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    ("ed_batch_fit_bins_form", C.c_int, [_vp]),
    ("ed_batch_n_samples", C.c_int64, [_vp]),
    ("ed_batch_fit_bins_n_unconverged", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_batch_run_bins", C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    ("ed_batch_phi_linear", C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    ("ed_batch_fit_cov", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    ("ed_batch_run_cov", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_double, _vp]),
    ("ed_batch_expected_cov", C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    ("ed_batch_run", C.c_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _vp]),
    ("ed_batch_set_fused", C.c_int, [_vp, C.c_int]),
    ("ed_batch_keep_loglik", C.c_int, [_vp, C.c_int]),
    ("ed_batch_n_emit_launches", C.c_int, [_vp]),
    ("ed_batch_loglik", _vp, [_vp]),
    ("ed_batch_path", _vp, [_vp]),
    ("ed_batch_calls", _vp, [_vp]),
    ("ed_batch_n_calls", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_batch_n_gsl_errors", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_batch_copy_calls", C.c_int, [_vp, _vp, _i64]),
    ("ed_batch_copy_call_info", C.c_int, [_vp, _vp, _i64]),
    ("ed_batch_copy_path", C.c_int, [_vp, _vp]),
    ("ed_batch_copy_loglik", C.c_int, [_vp, _vp]),
    ("ed_batch_verify_emissions", C.c_int, [_vp, _vp, _vp, _vp, _vp, _dbl, C.POINTER(_i64), C.POINTER(_i64), _vp, _i64]),
    ("ed_batch_enable_timing", C.c_int, [_vp, C.c_int]),
    ("ed_batch_stage_ms", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("ed_batch_stage_ms_total", C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i64)]),
    ("ed_batch_set_async_tail", C.c_int, [_vp, C.c_int]),
    ("ed_batch_set_viterbi_overlap", C.c_int, [_vp, C.c_int]),
    ("ed_batch_wait", C.c_int, [_vp, _vp]),
    ("ed_select_reference_set", C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, C.POINTER(_i32), C.POINTER(_i64), _vp]),
    ("ed_select_reference_set_part", C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, C.POINTER(_i32), C.POINTER(_i64), _vp]),
    ("ed_cohort_select_reference_sets", C.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    ("ed_cohort_select_reference_sets_range", C.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    ("ed_cohort_select_reference_sets_sm", C.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    ("ed_cohort_select_reference_sets_host", C.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    ("ed_release_scratch", C.c_int, []),
    ("ed_refcohort_last_path", C.c_int, [C.POINTER(C.c_int64)]),
    ("ed_refset_finalize", C.c_int, [_vp, _i64, C.POINTER(_i32)]),
    ("ed_refset_thin_positions", C.c_int, [_i64, _i64, _vp, _i64, C.POINTER(_i64)]),
    ("ed_get_power_betabinom", C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp]),
    ("ed_get_power_betabinom_mode", C.c_int, [_i64, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("ed_cohort_create", C.c_int, [C.POINTER(_vp), _vp, _i64, C.c_int]),
    ("ed_cohort_destroy", None, [_vp]),
    ("ed_cohort_set_option", C.c_int, [_vp, C.c_char_p, _dbl]),
    ("ed_cohort_submit", C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _dbl, _vp, C.POINTER(_i64)]),
    ("ed_cohort_batch", C.c_int, [_vp, _i64, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    ("ed_cohort_copy_params", C.c_int, [_vp, _i64, _vp, _vp]),
    ("ed_cohort_copy_bins_params", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    ("ed_cohort_copy_bins", C.c_int, [_vp, _vp, _vp]),
    ("ed_cohort_wait", C.c_int, [_vp, _i64]),
    ("ed_cohort_drain", C.c_int, [_vp]),
    ("ed_cohort_stream", _vp, [_vp]),
    ("ed_cohort_stage_ms_total", C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i64)]),
    ("ed_cohort_n_emit_launches", C.c_int, [_vp]),
    ("ed_cohort_emission_intervals", C.c_int, [_vp, _vp, _i64, C.POINTER(_i64)]),
    ("ed_cohort_submit_host", C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _i64, _vp, _vp, _dbl, C.POINTER(_i64)]),
    ("ed_cohort_submit_host_test", C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _i64, _vp, _vp, _dbl, C.POINTER(_i64)]),
    ("ed_cohort_ingest_stats", C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("ed_cohort_n_wide_slabs", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_host_alloc", C.c_int, [C.POINTER(_vp), C.c_size_t]),
    ("ed_host_free", C.c_int, [_vp]),
    ("ed_cohort_run_host", C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp, _vp, _dbl, _vp, _vp, _vp, C.POINTER(_i64)]),
    ("ed_cohort_copy_calls", C.c_int, [_vp, _vp, _vp, _i64]),
    ("ed_cohort_run_status", C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    ("ed_multi_create", C.c_int, [C.POINTER(_vp), _vp, C.c_int, _i64, _i32, _vp, _vp, _vp, _dbl, _dbl, _i64, C.c_int]),
    ("ed_multi_destroy", None, [_vp]),
    ("ed_multi_n_devices", C.c_int, [_vp]),
    ("ed_multi_set_option", C.c_int, [_vp, C.c_char_p, _dbl]),
    ("ed_multi_run_host", C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp, _vp, _dbl, _vp, _vp, _vp, C.POINTER(_i64)]),
    ("ed_multi_copy_calls", C.c_int, [_vp, _vp, _vp, _i64]),
    ("ed_multi_run_status", C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    ("ed_multi_table_status", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_multi_copy_bins", C.c_int, [_vp, _vp, _vp]),
    ("ed_multi_device_stats", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    ("ed_fit_betabin_host", C.c_int, [_vp, _vp, _i64, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    ("ed_select_reference_set_host", C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, C.POINTER(_i32), C.POINTER(_i64)]),
    ("ed_batch_set_fit_mode", C.c_int, [_vp, C.c_int]),
    ("ed_batch_set_emit_mode", C.c_int, [_vp, C.c_int]),
    ("ed_batch_set_counts_layout", C.c_int, [_vp, C.c_int]),
    ("ed_batch_set_counts_bits", C.c_int, [_vp, C.c_int]),
    ("ed_batch_set_emit_tables", C.c_int, [_vp, _i32, _i32, _dbl]),
    ("ed_batch_verify_emissions_tol", C.c_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, C.POINTER(_i64), C.POINTER(_i64),
                                               C.POINTER(_dbl), C.POINTER(_dbl), _vp, _i64]),
    ("ed_batch_copy_emit_tables", C.c_int, [_vp, _i64, C.POINTER(_i32), _vp, _i64]),
    ("ed_batch_n_cold_cells", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_batch_copy_table_dims", C.c_int, [_vp, _i64, C.POINTER(_i32)]),
    ("ed_batch_copy_table_windows", C.c_int, [_vp, _i64, C.POINTER(_i32)]),
    ("ed_batch_set_emit_tails", C.c_int, [_vp, C.c_int]),
    ("ed_batch_table_stats", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_cohort_table_status", C.c_int, [_vp, C.POINTER(_i64)]),
    ("ed_malloc", C.c_int, [C.POINTER(_vp), C.c_size_t]),
    ("ed_free", C.c_int, [_vp]),
    ("ed_memcpy_h2d", C.c_int, [_vp, _vp, C.c_size_t]),
    ("ed_memcpy_d2h", C.c_int, [_vp, _vp, C.c_size_t]),
    ("ed_synchronize", C.c_int, [_vp]),
    ("ed_eval_sf", C.c_int, [C.c_int, _i64, _vp, _vp, _vp]),
]


def lib():
    """Load libedcore.so; raise loudly if it is not there (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EdError("exomedepth_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != ED_OK:
        raise EdError("libedcore: %s (status %d)" % (lib().ed_last_error().decode(errors="replace"), rc))
