"""Two batches in flight (ed_batch_set_async_tail): same bits as the synchronous path, whatever the interleaving; plus
the round-1 advisor's findings on the per-sample mirror (prop.tumor in CallCNVs, run inputs kept alive across fit())."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_batches_in_flight_give_the_synchronous_results(edlib, oracle):
    torch = pytest.importorskip("torch")
    from exomedepth_amd import synth
    E, S, C = 30000, 200, 7
    dev = torch.device("cuda", 0)
    chrom_off, start, end = synth.exon_design(E, C, seed=5)
    plan = edlib.Plan(chrom_off, start, end)
    data = [synth.counts_torch(chrom_off, S, dev, seed=100 + k) for k in range(4)]      # four different cohorts
    # synchronous reference results, one batch after the other
    ref_out = []
    b0 = edlib.Batch(plan, S)
    for test, ref, p, phi in data:
        ph = torch.empty(S, dtype=torch.float64, device=dev); pe = torch.empty(S, dtype=torch.float64, device=dev)
        b0.fit(test, ref, ph, pe)
        b0.run(test, ref, ph, pe)
        ref_out.append((b0.path().copy(), b0.calls().copy(), b0.loglik().copy(), b0.call_info().copy(), ph.cpu().numpy(), pe.cpu().numpy()))
    b0.close()
    # pipelined: two batches used alternately, all emissions on one stream, fits on another, nothing synchronised in between
    batches = [edlib.Batch(plan, S) for _ in range(2)]
    for b in batches:
        b.set_async_tail(True)
        b.enable_timing(True)
    batches[0].set_viterbi_overlap(False)         # one launch of emissions, then all chains (bench.py's schedule)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    outs = [(torch.empty(S, dtype=torch.float64, device=dev), torch.empty(S, dtype=torch.float64, device=dev)) for _ in range(2)]
    got = []
    for rounds in range(3):                       # 12 steps: every batch object is reused while its predecessor's tail is in flight
        for i, (test, ref, p, phi) in enumerate(data):
            b = batches[i % 2]
            ph, pe = outs[i % 2]
            if i >= 2 or rounds:                  # results of the run two steps ago on this batch object (synchronises that batch only)
                got.append((b.path().copy(), b.calls().copy(), b.loglik().copy(), b.call_info().copy(), ph.cpu().numpy(), pe.cpu().numpy()))
            b.fit(test, ref, ph, pe, stream=side.cuda_stream)
            main.wait_stream(side)
            b.run(test, ref, ph, pe, stream=main.cuda_stream)
    for i in (2, 3):
        b = batches[i % 2]
        got.append((b.path().copy(), b.calls().copy(), b.loglik().copy(), b.call_info().copy(), outs[i % 2][0].cpu().numpy(), outs[i % 2][1].cpu().numpy()))
    assert len(got) == 12
    for k, g in enumerate(got):
        want = ref_out[k % 4]
        for a, w in zip(g, want):
            assert a.tobytes() == w.tobytes(), ("step", k)
    tot, nr, nf = batches[0].stage_ms_total()
    assert nr == 6 and nf == 6 and tot["emissions"] > 0
    # ed_batch_wait: another stream can be made to wait for the tail on the device
    third = torch.cuda.Stream()
    batches[1].wait(third.cuda_stream)
    third.synchronize()
    for b in batches:
        b.close()
    plan.close()


def test_callcnvs_with_prop_tumor_runs_viterbi_on_the_likelihood_slot(edlib, oracle):
    """reference R/class_definition.R:364 runs the Viterbi on x@likelihood, which new('ExomeDepth', prop.tumor = m) built with
    mixture m (:184-189).  (the mirror used mixture 1 for the path and m for BF.)"""
    import io, contextlib
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "exomecount_chr1.npz"))
    counts = g["counts"][:6000]
    test = counts[:, 0].astype(float); ref = counts[:, 1:].sum(axis=1).astype(float)
    start, end = g["start"][:6000], g["end"][:6000]
    chrom = ["1"] * len(start)
    with contextlib.redirect_stdout(io.StringIO()):
        x = edlib.ExomeDepth(test, ref, phi=0.0049568, expected=0.2030, prop_tumor=0.3)
        x.CallCNVs(chrom, start, end, ["e%d" % i for i in range(len(start))])
    L, _ = oracle.get_loglike_matrix(0.0049568, 0.2030, (test + ref).astype(np.int32), test.astype(np.int32), 0.3, oracle.PORTABLE)
    assert np.array_equal(x.likelihood.view(np.int64), np.ascontiguousarray(L).view(np.int64))
    epath, ecalls = oracle.callcnvs(L, np.array([0, len(start)], np.int32), start, end)
    assert np.array_equal(x.Viterbi_path, epath)
    assert [c["start.p"] for c in x.CNV_calls] == ecalls[:, 0].astype(int).tolist()
    L1, _ = oracle.get_loglike_matrix(0.0049568, 0.2030, (test + ref).astype(np.int32), test.astype(np.int32), 1.0, oracle.PORTABLE)
    p1, _ = oracle.callcnvs(L1, np.array([0, len(start)], np.int32), start, end)
    assert not np.array_equal(p1, epath)          # the mixture does change the path on this data: the test can tell


def test_call_info_after_a_fit_reads_live_inputs(edlib):
    """run(host arrays) -> fit(...) -> call_info() must not read freed device temporaries."""
    from exomedepth_amd import synth
    E, S, C = 3000, 40, 3
    chrom_off, start, end = synth.exon_design(E, C, 2)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 2, n_segments=4, mean_depth=80.0)
    plan = edlib.Plan(chrom_off, start, end)
    b = edlib.Batch(plan, S)
    b.run(test, ref, phi, p)
    want = b.call_info().copy()
    dphi = edlib.DeviceArray(np.zeros(S)); dexp = edlib.DeviceArray(np.zeros(S))
    b.fit(test, ref, dphi, dexp)
    junk = [edlib.DeviceArray(np.full((E, S), 7, np.int32)) for _ in range(4)]      # would recycle freed blocks
    assert b.call_info().tobytes() == want.tobytes() and len(want) > 0
    del junk
    b.close(); plan.close()


def test_fit_reports_convergence_and_plan_rejects_unrepresentable_padding(edlib):
    from exomedepth_amd import synth
    E, S, C = 4000, 30, 2
    chrom_off, start, end = synth.exon_design(E, C, 4)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 4, n_segments=2, mean_depth=70.0)
    plan = edlib.Plan(chrom_off, start, end)
    b = edlib.Batch(plan, S)
    dphi = edlib.DeviceArray(np.zeros(S)); dexp = edlib.DeviceArray(np.zeros(S))
    with pytest.raises(edlib.EdError, match="no ed_batch_fit"):
        b.fit_unconverged()
    for mode in (1, 0):
        b.set_fit_histograms(mode)
        b.fit(test, ref, dphi, dexp)
        assert b.fit_unconverged() == (0, -1)
    b.close(); plan.close()
    # as.integer(start - 2 L) would be NA in R (R/class_definition.R:368): an error here, not undefined behaviour
    far = start.astype(np.int64) + (2**31 - 1 - int(end.max()) - 10)
    with pytest.raises(edlib.EdError, match="does not fit a 32-bit integer"):
        edlib.Plan(chrom_off, far.astype(np.int32), (far + (end - start)).astype(np.int32), 1e-4, 50000.0)


def test_device_call_table_is_the_call_table(edlib):
    """dist.device_call_table: the library's device-resident table wrapped as a torch tensor (what the RCCL gather sends)."""
    torch = pytest.importorskip("torch")
    from exomedepth_amd import synth, dist as eddist, api
    E, S, C = 5000, 70, 4
    chrom_off, start, end = synth.exon_design(E, C, 6)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, 6, n_segments=5, mean_depth=90.0)
    plan = edlib.Plan(chrom_off, start, end)
    b = edlib.Batch(plan, S)
    b.run(test, ref, phi, p)
    t = eddist.device_call_table(b)
    calls = b.calls()
    assert t.is_cuda and t.dtype == torch.int32 and tuple(t.shape) == (len(calls), 6) and len(calls) > 0
    assert np.array_equal(t.cpu().numpy(), np.ascontiguousarray(calls).view(np.int32).reshape(-1, 6))
    assert np.array_equal(eddist.calls_to_tensor(calls, torch.device("cpu")).numpy(), t.cpu().numpy())
    b.close(); plan.close()
