"""Depth-binned dispersion (`phi.bins > 1`, reference R/class_definition.R:120-147) through the C-ABI.

Parity status.  Binning, interpolation and emissions are restated from the R/C sources and are compared exactly
(complete.bins and phi.linear bit for bit against oracle/bins_oracle.py; the likelihood bit for bit against the
checker's get_loglike_matrix fed the same per-exon phi).  The per-level dispersions come from aod::betabin in the
reference (not in the reference tree): UNPINNED, compared at FIT_REL_TOL with the checker's long-double MLE of the
documented likelihood.
"""
import numpy as np
import pytest

from test_gpu_parity import bits

pytestmark = pytest.mark.gpu

FIT_REL_TOL = 1e-7


def _case(E, S, C, seed, depth=90.0):
    from exomedepth_amd import synth
    chrom_off, start, end = synth.exon_design(E, C, seed)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=3, mean_depth=depth)
    return chrom_off, start, end, test, ref


@pytest.mark.parametrize("B", [2, 3, 5])
def test_fit_bins_matches_checker(edlib, oracle, B):
    from oracle import bins_oracle as bo
    E, S = 5000, 6
    chrom_off, start, end, test, ref = _case(E, S, 3, 70 + B)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dphib = edlib.DeviceArray(np.zeros((B, S)))
    dedges = edlib.DeviceArray(np.zeros((B + 1, S)))
    dexp = edlib.DeviceArray(np.zeros(S))
    batch.fit_bins(test, ref, B, dphib, dedges, dexp)
    phib, edges, exp = dphib.to_host(), dedges.to_host(), dexp.to_host()
    philin = batch.phi_linear(ref, B, dphib, dedges)
    batch.close(); plan.close()
    for s in range(S):
        ophi, op, olin, ocomplete = bo.fit_bins(test[:, s], ref[:, s], B)
        assert np.array_equal(bits(edges[:, s]), bits(ocomplete)), (s, edges[:, s], ocomplete)
        assert np.max(np.abs(phib[:, s] - ophi) / ophi) < FIT_REL_TOL, (s, phib[:, s], ophi)
        assert abs(exp[s] - op) / op < FIT_REL_TOL
        # the interpolation itself, fed the device's own estimates: exact
        mid = (edges[:B, s] + edges[1:B + 1, s]) / 2
        mine = bo.approx_linear(ref[:, s].astype(np.float64), mid, phib[:, s])
        assert np.array_equal(bits(philin[:, s]), bits(mine))


def test_binning_failure_is_reported(edlib):
    """a constant reference column puts every exon in the top level: the reference stops with
    'Binning did not happen properly' (R/class_definition.R:130-133)"""
    E, S, B = 600, 3, 3
    chrom_off, start, end, test, ref = _case(E, S, 2, 90)
    ref[:, 1] = 500
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    d = [edlib.DeviceArray(np.zeros((B, S))), edlib.DeviceArray(np.zeros((B + 1, S))), edlib.DeviceArray(np.zeros(S))]
    with pytest.raises(edlib.EdError, match="Binning did not happen properly"):
        batch.fit_bins(test, ref, B, *d)
    with pytest.raises(edlib.EdError):
        batch.fit_bins(test, ref, 1, *d)        # phi.bins = 1 is the plain fit
    batch.close(); plan.close()


def test_run_bins_pipeline_parity(edlib, oracle):
    """fit -> per-exon phi -> emissions -> Viterbi -> calls; the checker, fed the device's phi.linear, must agree
    bit for bit on the likelihood, the path and the call table."""
    E, S, C, B = 4000, 5, 4, 4
    chrom_off, start, end, test, ref = _case(E, S, C, 95, depth=60.0)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dphib = edlib.DeviceArray(np.zeros((B, S)))
    dedges = edlib.DeviceArray(np.zeros((B + 1, S)))
    dexp = edlib.DeviceArray(np.zeros(S))
    batch.fit_bins(test, ref, B, dphib, dedges, dexp)
    batch.run_bins(test, ref, B, dphib, dedges, dexp)
    ll, path, calls = batch.loglik(), batch.path(), batch.calls()
    info = batch.call_info()
    philin = batch.phi_linear(ref, B, dphib, dedges)
    exp = dexp.to_host()
    batch.set_fused(True)
    with pytest.raises(edlib.EdError):
        batch.run_bins(test, ref, B, dphib, dedges, dexp)
    batch.close(); plan.close()
    assert len(info) == len(calls)
    k = 0
    for s in range(S):
        exp_ll, _ = oracle.get_loglike_matrix(philin[:, s], np.full(E, exp[s]), test[:, s] + ref[:, s], test[:, s], 1.0,
                                              oracle.PORTABLE)
        assert np.array_equal(bits(ll[:, :, s]), bits(exp_ll)), "sample %d" % s
        # the .Call-shaped entry with the same per-exon phi gives the same bits
        mine = np.array(edlib.get_loglike_matrix(philin[:, s], np.full(E, exp[s]), test[:, s] + ref[:, s], test[:, s]))
        assert np.array_equal(bits(mine), bits(exp_ll))
        exp_path, exp_calls = oracle.callcnvs(exp_ll, chrom_off, start, end)
        assert np.array_equal(path[:, s].astype(np.int8), exp_path)
        m = calls[calls["sample"] == s]
        assert len(m) == len(exp_calls)
        assert np.array_equal(m["start_exon"] + 1, exp_calls[:, 0].astype(np.int64))
        assert np.array_equal(m["end_exon"] + 1, exp_calls[:, 1].astype(np.int64))
        assert np.array_equal(m["type"], exp_calls[:, 2].astype(np.int64))
        k += len(m)
    assert k == len(calls) and k > 0


def test_mirror_phi_bins(edlib, oracle):
    """ExomeDepth(..., phi_bins = B).CallCNVs(...) -- the per-sample mirror of the S4 flow with variable phi."""
    from oracle import bins_oracle as bo
    E, C, B = 3000, 3, 3
    chrom_off, start, end, test, ref = _case(E, 2, C, 97)
    t, r = test[:, 0].astype(float), ref[:, 0].astype(float)
    x = edlib.ExomeDepth(t, r, phi_bins=B)
    ophi, op, olin, _ = bo.fit_bins(test[:, 0], ref[:, 0], B)
    assert x.phi.shape == (E,) and np.max(np.abs(x.phi - olin) / olin) < FIT_REL_TOL
    assert abs(x.expected[0] - op) / op < FIT_REL_TOL
    chrom = np.concatenate([[str(c + 1)] * int(chrom_off[c + 1] - chrom_off[c]) for c in range(C)])
    x.CallCNVs(chrom, start, end, np.array(["e%d" % i for i in range(E)]))
    exp_ll, _ = oracle.get_loglike_matrix(x.phi, x.expected, test[:, 0] + ref[:, 0], test[:, 0], 1.0, oracle.PORTABLE)
    assert np.array_equal(bits(x.likelihood), bits(exp_ll))
    exp_path, exp_calls = oracle.callcnvs(exp_ll, chrom_off, start, end)
    assert np.array_equal(x.Viterbi_path.astype(np.int8), exp_path)
    assert [c["start.p"] for c in x.CNV_calls] == list(exp_calls[:, 0].astype(int))
    assert [c["end.p"] for c in x.CNV_calls] == list(exp_calls[:, 1].astype(int))
    with pytest.raises(ValueError):
        edlib.ExomeDepth(t, r, phi_bins=B, subset_for_speed=100)


def test_fit_bins_few_exons_per_level(edlib, oracle):
    """949 exons in 7 depth levels at low depth: a case of tools/fuzz_more.py on which the grouped Newton iteration
    without step control ended 20 % off in p (it now starts from the single-dispersion optimum, shifts a non-concave
    Hessian, scales the step as a whole and backs off after an overshoot)."""
    from exomedepth_amd import synth
    from oracle import bins_oracle as bo
    E, S, B, seed = 949, 1, 7, 309133923
    chrom_off, start, end = synth.exon_design(E, 1, seed)
    test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=30.0)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    dphib = edlib.DeviceArray(np.zeros((B, S))); dedges = edlib.DeviceArray(np.zeros((B + 1, S))); dexp = edlib.DeviceArray(np.zeros(S))
    batch.fit_bins(test, ref, B, dphib, dedges, dexp)
    phib, exp = dphib.to_host(), dexp.to_host()
    batch.close(); plan.close()
    ophi, op, _, _ = bo.fit_bins(test[:, 0], ref[:, 0], B)
    assert np.max(np.abs(phib[:, 0] - ophi) / ophi) < 1e-6, (phib[:, 0], ophi)
    assert abs(exp[0] - op) / op < 1e-7


@pytest.mark.parametrize("B,E,S,depth", [(3, 60000, 21, 90.0), (4, 40000, 9, 150.0), (8, 70001, 5, 60.0), (2, 3000, 3, 40.0)])
def test_histogram_form_equals_per_cell_form(edlib, B, E, S, depth):
    """ed_batch_fit_bins on count histograms (csrc/edbins_hist.inc: r bins for the quantile, per-level y and n bins, listed
    cells) against the per-cell form: complete.bins bit for bit, the estimates to the fit's tolerance"""
    chrom_off, start, end, test, ref = _case(E, S, 4, 500 + B, depth=depth)
    plan = edlib.Plan(chrom_off, start, end)
    out = []
    for form in (1, 0):
        batch = edlib.Batch(plan, S)
        batch.set_fit_histograms(form)
        d = [edlib.DeviceArray(np.zeros((B, S))), edlib.DeviceArray(np.zeros((B + 1, S))), edlib.DeviceArray(np.zeros(S))]
        for rep in range(2):                      # the second fit reuses the first one's buffers
            batch.fit_bins(test, ref, B, *d)
            assert batch.fit_bins_form == form
        out.append([x.to_host() for x in d])
        batch.close()
    plan.close()
    (phib, edges, exp), (phib0, edges0, exp0) = out
    assert np.array_equal(bits(edges), bits(edges0))
    assert np.max(np.abs(phib - phib0) / phib0) < FIT_REL_TOL
    assert np.max(np.abs(exp - exp0) / exp0) < FIT_REL_TOL


def test_data_beyond_the_bins_take_the_per_cell_form(edlib, oracle):
    """reference counts whose 0.85 quantile lies beyond the 8192 unit bins: the histogram form declines, the answer is the checker's"""
    from oracle import bins_oracle as bo
    E, S, B = 4000, 3, 3
    chrom_off, start, end, test, ref = _case(E, S, 2, 77, depth=120.0)
    ref = ref * 40
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    d = [edlib.DeviceArray(np.zeros((B, S))), edlib.DeviceArray(np.zeros((B + 1, S))), edlib.DeviceArray(np.zeros(S))]
    batch.fit_bins(test, ref, B, *d)
    assert batch.fit_bins_form == 0
    phib, edges, exp = [x.to_host() for x in d]
    batch.close(); plan.close()
    for s in range(S):
        ophi, op, olin, ocomplete = bo.fit_bins(test[:, s], ref[:, s], B)
        assert np.array_equal(bits(edges[:, s]), bits(ocomplete))
        assert np.max(np.abs(phib[:, s] - ophi) / ophi) < FIT_REL_TOL


@pytest.mark.parametrize("E,S,seed,col", [(806, 8, 271408301, 4), (806, 8, 271408301, 1), (1741, 3, 861751877, 1)])
@pytest.mark.parametrize("form", [1, 0])
def test_an_under_dispersed_level_is_held_on_the_floor(edlib, oracle, E, S, seed, col, form):
    """Cases of tools/fuzz_bins.py (8 levels at ~12 reads per exon): the lowest depth level is under-dispersed, its maximum-likelihood
    dispersion is 0 (the checker follows it to ~1e-10, the device holds it on its floor 1e-6).  Kept inside the coupled Newton system, that
    level made every step a shifted one and the 40 passes ended 1e-3 short in the OTHER levels; the overshoot test then halved a
    displacement that the bound had bent downhill for 20 passes.  Both forms now reach the checker's values in the free levels."""
    from exomedepth_amd import synth
    from oracle import bins_oracle as bo
    B = 8
    chrom_off, start, end = synth.exon_design(E, 1, seed)
    test, ref, _, _, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=12.0)
    plan = edlib.Plan(chrom_off, start, end)
    batch = edlib.Batch(plan, S)
    batch.set_fit_histograms(form)
    d = [edlib.DeviceArray(np.zeros((B, S))), edlib.DeviceArray(np.zeros((B + 1, S))), edlib.DeviceArray(np.zeros(S))]
    batch.fit_bins(test, ref, B, *d)
    assert batch.fit_bins_form == form
    phib, exp = d[0].to_host()[:, col], d[2].to_host()[col]
    batch.close(); plan.close()
    ophi, op, _, _ = bo.fit_bins(test[:, col], ref[:, col], B)
    assert ophi[0] < 1e-8 and phib[0] <= 1.0000001e-6
    assert np.max(np.abs(phib[1:] - ophi[1:]) / ophi[1:]) < 2e-5, (phib, ophi)
    assert abs(exp - op) / op < 2e-5
