"""exomedepth_amd -- MI355X (gfx950) CNV-calling core: beta-binomial emissions + 3-state Viterbi.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C-ABI of include/exomedepth_amd.h)
and the host-side mirror of the reference's interface for this path (api.py).
"""
from ._lib import EdError, LIB_PATH  # noqa: F401
from .api import (Batch, Cohort, MultiDevice, DeviceArray, device_count, PinnedArray, ExomeDepth, Plan, chromosome_order, cohort_select_reference_sets, refcohort_last_path, fit_betabin, fit_betabin_bins,  # noqa: F401
                  get_loglike_matrix, get_power_betabinom, refset_finalize, select_reference_set, viterbi_hmm)

__all__ = ["Batch", "Cohort", "MultiDevice", "DeviceArray", "device_count", "PinnedArray", "ExomeDepth", "Plan", "EdError", "chromosome_order", "cohort_select_reference_sets", "refcohort_last_path", "fit_betabin", "fit_betabin_bins",
           "get_loglike_matrix", "get_power_betabinom", "refset_finalize", "select_reference_set", "viterbi_hmm", "LIB_PATH"]
