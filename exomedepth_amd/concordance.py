"""How much the choice of estimator for (phi, expected) moves the results of the path.

Emissions, Viterbi paths and call tables are bit-pinned against the reference GIVEN (phi, expected).  The parameters themselves
come from a fit the reference delegates to aod::betabin (R/class_definition.R:118: Nelder-Mead on the beta-binomial
likelihood, optim()'s reltol 1.5e-8 on the objective), which stops within ~1e-3 of the maximum in phi.  The library offers both
ends: fit mode 0, the maximum-likelihood estimate itself (Newton, 1e-9), and fit mode 1, aod's own procedure (csrc/edfit_hist.inc
k_fit_hnm).  This module runs the whole path once with each and counts what differs -- the number bench.py reports as
`fit_concordance` and tests/test_gpu_fit_concordance.py pins.  Product code: no oracle involved, both fits run on the device.
"""
import numpy as np

from . import api
from ._lib import check, lib


def _run_with_mode(plan, test, ref, mode, mixture):
    S = int(test.shape[1])
    b = api.Batch(plan, S)
    try:
        check(lib().ed_batch_set_fit_mode(b.handle, int(mode)))
        dphi, dexp = api.DeviceArray(np.zeros(S)), api.DeviceArray(np.zeros(S))
        b.fit(test, ref, dphi, dexp)
        b.run(test, ref, dphi, dexp, mixture)
        n_unconv = b.fit_unconverged()[0]
        return {"phi": dphi.to_host(), "expected": dexp.to_host(), "path": b.path(), "calls": b.calls(), "loglik": b.loglik(),
                "unconverged": n_unconv}
    finally:
        b.close()


def fit_mode_concordance(plan, test, ref, mixture=1.0):
    """test, ref: (n_exons, S) int32 host arrays (or device tensors).  Returns a dict of plain numbers."""
    r0 = _run_with_mode(plan, test, ref, 0, mixture)      # maximum likelihood
    r1 = _run_with_mode(plan, test, ref, 1, mixture)      # aod-nm
    E, S = r0["path"].shape
    l0, l1 = r0["loglik"], r1["loglik"]
    ok = np.isfinite(l0) & np.isfinite(l1) & (l0 != 0)
    rel = np.abs(l0[ok] - l1[ok]) / np.abs(l0[ok])
    c0 = {tuple(int(v) for v in row) for row in r0["calls"]}
    c1 = {tuple(int(v) for v in row) for row in r1["calls"]}
    dphi = np.abs(r0["phi"] - r1["phi"]) / r0["phi"]
    dp = np.abs(r0["expected"] - r1["expected"]) / r0["expected"]
    diff = r0["path"] != r1["path"]
    return {"columns": int(S), "exons": int(E), "cells": int(E) * int(S),
            "max_rel_dphi": float(dphi.max()), "median_rel_dphi": float(np.median(dphi)),
            "max_rel_dexpected": float(dp.max()),
            "max_rel_dloglik": float(rel.max()) if rel.size else 0.0,
            "discordant_states": int(diff.sum()), "columns_with_discordant_states": int(diff.any(axis=0).sum()),
            "calls_mle": len(c0), "calls_aod_nm": len(c1), "discordant_call_rows": len(c0 ^ c1),
            "unconverged_mle": int(r0["unconverged"]), "unconverged_aod_nm": int(r1["unconverged"]),
            "what": "whole path run twice on the same counts: (phi, expected) from the device's maximum-likelihood fit vs from aod::betabin's "
                    "Nelder-Mead procedure restated on the device (fit mode 1); aod itself is not in the reference tree (parity unpinned)"}
