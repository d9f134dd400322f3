"""Time-bounded slice of tools/soak_emission.py: every cell of randomly shaped batches evaluated twice on the device --
k_emit_batch (hoisted constants, tables, route binning, overlapped with the Viterbi kernels) against the reference's
per-cell loop (k_emit_verify: six log-Betas per cell, src/CNV_estimate.cpp:71-81) -- bits compared on the device."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_dual_evaluation_soak_small_shapes_and_exome_depths(edlib):
    pytest.importorskip("torch")
    import soak_emission
    res = soak_emission.soak(seconds=20.0, seed=20250928, log=print)
    assert res["total"]["mismatch"] == 0, res["events"]
    assert res["total"]["cells"] > 2e9, res["total"]            # ~20 s at a few 1e9 cells/s
    for regime in ("small", "tiny", "exome"):
        assert res["per_regime"][regime]["runs"] > 0


def test_verify_emissions_detects_a_flipped_bit(edlib, oracle):
    """The self-check is not vacuous: it agrees with the CPU checker on a clean run and reports exactly the cell whose
    likelihood was computed from different inputs."""
    from exomedepth_amd import synth
    E, S, C = 600, 70, 3
    chrom_off, start, end = synth.exon_design(E, C, 3)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 3, n_segments=2, mean_depth=8.0)
    phi = np.minimum(phi * 40.0, 0.5)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    batch.run(test, ref, phi, p)
    ncmp, nbad, first = batch.verify_emissions(test, ref, phi, p)
    assert (ncmp, nbad, first) == (3 * E * S, 0, [])
    ll = batch.loglik()
    for s in (0, 33, 69):
        ell, _ = oracle.get_loglike_matrix(phi[s], p[s], test[:, s] + ref[:, s], test[:, s], 1.0, oracle.PORTABLE)
        assert np.array_equal(ll[:, :, s].view(np.int64), np.ascontiguousarray(ell).view(np.int64))
    # verify against inputs that differ in one cell: that cell's values (and only those) are reported
    t2 = test.copy()
    t2[123, 45] += 1
    ncmp, nbad, first = batch.verify_emissions(t2, ref, phi, p)
    assert ncmp == 3 * E * S and 1 <= nbad <= 3
    assert all(m["exon"] == 123 and m["sample"] == 45 for m in first) and len(first) == nbad
    batch.close(); plan.close()
