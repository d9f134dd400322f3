"""The cohort pipeline of the library (ed_cohort_*: own streams, batch rotation, event-ordered stages, ingest from host
memory): the same bits as ed_batch_fit + ed_batch_run slab by slab, whatever the options, layouts and wire formats.
Counterpart in the reference: the user's loop over samples, vignette/vignette.Rnw:390-431."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference_results(edlib, plan, slabs, emit_mode=0):
    """slab by slab, synchronously, through the batch interface"""
    out = []
    for test, ref in slabs:
        n = test.shape[1]
        b = edlib.Batch(plan, n)
        if emit_mode:
            b.set_emit_mode(emit_mode)
        dphi = edlib.DeviceArray(np.zeros(n)); dexp = edlib.DeviceArray(np.zeros(n))
        b.fit(test, ref, dphi, dexp)
        b.run(test, ref, dphi, dexp)
        out.append({"calls": b.calls().copy(), "info": b.call_info().copy(), "path": b.path().copy(), "loglik": b.loglik().copy(),
                    "phi": dphi.to_host(), "expected": dexp.to_host()})
        b.close()
    return out


_CACHE = {}


def _same(got, want, keys):
    bad = [k for k in keys if got[k].tobytes() != want[k].tobytes()]
    detail = []
    for k in bad:                                                  # (what differs, and by how much: a race shows up as a few low bits or as another slab's values)
        g, w = np.asarray(got[k]), np.asarray(want[k])
        if g.dtype.names:
            detail.append((k, [n for n in g.dtype.names if g[n].tobytes() != w[n].tobytes()], int(sum(np.sum(g[n] != w[n]) for n in g.dtype.names))))
        elif g.shape == w.shape:
            d = np.abs(g.astype(np.float64) - w.astype(np.float64))
            detail.append((k, int(np.sum(g != w)), float(np.nanmax(d))))
    assert not bad, detail


@pytest.fixture(scope="module")
def cohort_data(edlib):
    from exomedepth_amd import synth
    E, C, S = 12000, 5, 160
    chrom_off, start, end = synth.exon_design(E, C, seed=11)
    slabs = []
    for k, n in enumerate((S, S, S, S, 70)):                       # the last slab is ragged
        test, ref, _, _, _ = synth.counts_numpy(chrom_off, n, seed=300 + k, n_segments=4, mean_depth=80.0)
        slabs.append((test, ref))
    plan = edlib.Plan(chrom_off, start, end)
    want = _reference_results(edlib, plan, slabs)
    yield edlib, plan, slabs, want, S
    plan.close()


def test_lanes_option_is_validated(cohort_data):
    edlib, plan, slabs, want, S = cohort_data
    co = edlib.Cohort(plan, S, 4)
    with pytest.raises(edlib.EdError, match="lanes"):
        co.set_option("lanes", 3)                                  # not a divisor of the four slabs in flight
    with pytest.raises(edlib.EdError, match="lanes"):
        co.set_option("lanes", 5)
    co.set_option("lanes", 2)
    co.close()


# (option lanes: independent pipelines inside the object, slot s in lane s % lanes; 0 = slabs in flight / 2 lanes of two slots when that is 4, 6 or 8;
#  1, the default: the one pipeline the object was before)
@pytest.mark.parametrize("opts", [dict(), dict(own_queues=0), dict(split=0.0), dict(split=0.6, viterbi_overlap=1), dict(own_queues=0, split=0.5),
                                  dict(lanes=0), dict(lanes=0, own_queues=0), dict(emit_mode=2, lanes=0), dict(emit_mode=2), dict(lanes=2)])
@pytest.mark.parametrize("in_flight", [1, 2, 3, 4, 6, 8])
def test_device_slabs_through_the_cohort_equal_the_batch_interface(cohort_data, opts, in_flight):
    if opts.get("lanes", 1) > 1 and in_flight % opts["lanes"]:
        pytest.skip("lanes must divide the slabs in flight")
    edlib, plan, slabs, want, S = cohort_data
    if opts.get("emit_mode", 0):                                   # the same mode through the batch interface: the cohort must give its bits
        key = "want_mode_%d" % opts["emit_mode"]
        if key not in _CACHE:
            _CACHE[key] = _reference_results(edlib, plan, slabs, opts["emit_mode"])
        want = _CACHE[key]
    co = edlib.Cohort(plan, S, in_flight, timing=1, **opts)
    dev = [(edlib.DeviceArray(t), edlib.DeviceArray(r)) for t, r in slabs]
    tickets = []
    for rounds in range(2):                                        # every slot is reused while its predecessor is in flight
        for i, (dt, dr) in enumerate(dev):
            n = slabs[i][0].shape[1]
            if len(tickets) >= in_flight:                          # results of the ticket whose slot is about to be reused
                j = len(tickets) - in_flight
                got = co.results(tickets[j], slabs[j % len(slabs)][0].shape[1], path=True, loglik=True)
                _same(got, want[j % len(slabs)], ("calls", "info", "path", "loglik", "phi", "expected"))
            tickets.append(co.submit(dt, dr, n_samples=n))
    for j in range(len(tickets) - in_flight, len(tickets)):
        got = co.results(tickets[j], slabs[j % len(slabs)][0].shape[1], path=True, loglik=True)
        _same(got, want[j % len(slabs)], ("calls", "info", "path", "loglik", "phi", "expected"))
    with pytest.raises(edlib.EdError):                             # overwritten long ago
        co.batch(tickets[0])
    tot, nr, nf = co.stage_ms_total()
    assert nr == len(tickets) and nf == len(tickets) and tot["emissions"] > 0
    co.close()


@pytest.mark.parametrize("given", [False, True])
@pytest.mark.parametrize("opts", [dict(), dict(emit_mode=2), dict(lanes=2)])
def test_counts_produced_on_the_cohorts_own_stream(cohort_data, opts, given):
    """ready_stream = ed_cohort_stream() (lane 0's emission stream, which the header invites callers to produce on) with slabs in flight: the fit
    stream -- the fit, or for given parameters the constants / table statistics / table build -- reads the counts too and must be ordered behind
    the producer (the event was skipped when ready_stream was the emission stream itself).  The producer is made slow on purpose:
    a spin kernel, then the copy that fills the slab's buffers (zeros until then)."""
    torch = pytest.importorskip("torch")
    edlib, plan, slabs, want, S = cohort_data
    em = opts.get("emit_mode", 0)
    if em:
        key = "want_mode_%d" % em
        if key not in _CACHE:
            _CACHE[key] = _reference_results(edlib, plan, slabs, em)
        want = _CACHE[key]
    dev = torch.device("cuda", 0)
    co = edlib.Cohort(plan, S, 4, **opts)
    ext = torch.cuda.ExternalStream(co.stream, device=dev)
    src = [(torch.from_numpy(t).to(dev), torch.from_numpy(r).to(dev)) for t, r in slabs[:4]]
    par = [(torch.from_numpy(w["phi"]).to(dev), torch.from_numpy(w["expected"]).to(dev)) for w in want[:4]]
    torch.cuda.synchronize()
    tickets, keep = [], []
    for rnd in range(2):
        for i in range(4):
            dt, dr = torch.zeros_like(src[i][0]), torch.zeros_like(src[i][1])
            dp, de = torch.zeros_like(par[i][0]), torch.zeros_like(par[i][1])
            keep.append((dt, dr, dp, de))
            if len(tickets) >= 4:
                j = len(tickets) - 4
                got = co.results(tickets[j], S, path=True)
                _same(got, want[j % 4], ("calls", "path") if given else ("calls", "path", "phi", "expected"))
            with torch.cuda.stream(ext):
                torch.cuda._sleep(20_000_000)                      # ~10 ms: the counts are NOT there when the submission returns
                dt.copy_(src[i][0]); dr.copy_(src[i][1]); dp.copy_(par[i][0]); de.copy_(par[i][1])
            tickets.append(co.submit(dt, dr, phi=dp if given else None, expected=de if given else None, ready_stream=co.stream))
    for j in range(4, 8):
        got = co.results(tickets[j], S, path=True)
        _same(got, want[j % 4], ("calls", "path") if given else ("calls", "path", "phi", "expected"))
    co.close()


def test_given_parameters_and_mixture(cohort_data, oracle):
    edlib, plan, slabs, want, S = cohort_data
    test, ref = slabs[1]
    rng = np.random.default_rng(3)
    phi = rng.uniform(0.002, 0.02, S); p = rng.uniform(0.08, 0.2, S)
    b = edlib.Batch(plan, S)
    b.run(test, ref, phi, p, mixture=0.6)
    ref_calls, ref_path = b.calls().copy(), b.path().copy()
    b.close()
    co = edlib.Cohort(plan, S, 2)
    t0 = co.submit(edlib.DeviceArray(test), edlib.DeviceArray(ref), phi=edlib.DeviceArray(phi), expected=edlib.DeviceArray(p), mixture=0.6)
    got = co.results(t0, S, path=True)
    assert got["calls"].tobytes() == ref_calls.tobytes() and got["path"].tobytes() == ref_path.tobytes()
    assert got["phi"].tobytes() == phi.tobytes() and got["expected"].tobytes() == p.tobytes()
    co.close()


@pytest.mark.parametrize("layout,wire,pinned", [(0, 4, False), (0, 2, False), (1, 4, False), (1, 2, False), (0, 4, True), (1, 2, True), (0, 2, True)])
def test_whole_cohort_from_host_memory(cohort_data, layout, wire, pinned):
    """ed_cohort_run_host: the cohort's matrix in host memory -- sample-minor or R's column-major, int32 or the 16-bit wire
    format, pageable (staged through the pinned double buffer) or pinned (read in place) -- cut into slabs of 160 + 160 + ... + 70."""
    edlib, plan, slabs, want, S = cohort_data
    E = plan.n_exons
    test = np.concatenate([t for t, _ in slabs], axis=1)           # (E, 710)
    ref = np.concatenate([r for _, r in slabs], axis=1)
    St = test.shape[1]
    dt = np.int32 if wire == 4 else np.uint16
    assert test.max() < 65536 and ref.max() < 65536
    th = test.astype(dt) if layout == 0 else np.ascontiguousarray(test.T.astype(dt))
    rh = ref.astype(dt) if layout == 0 else np.ascontiguousarray(ref.T.astype(dt))
    keep = []
    if pinned:
        pt, pr = edlib.PinnedArray(th.shape, dt), edlib.PinnedArray(rh.shape, dt)
        pt.array[...] = th; pr.array[...] = rh
        th, rh = pt.array, pr.array
        keep = [pt, pr]
    co = edlib.Cohort(plan, S, 2)
    for rep in range(2):                                           # the second pass reuses every buffer of the first
        out = co.run_host(th, rh, layout, want_path=True)
        calls = np.concatenate([w["calls"] for w in want])
        shift = np.concatenate([np.full(len(w["calls"]), k * S, dtype=np.int32) for k, w in enumerate(want)])
        calls["sample"] += shift
        assert out["calls"].tobytes() == calls.tobytes()
        assert out["info"].tobytes() == np.concatenate([w["info"] for w in want]).tobytes()
        assert out["phi"].tobytes() == np.concatenate([w["phi"] for w in want]).tobytes()
        assert out["expected"].tobytes() == np.concatenate([w["expected"] for w in want]).tobytes()
        path = np.concatenate([w["path"] for w in want], axis=1)
        assert np.array_equal(out["path"] if layout == 0 else out["path"].T, path)
        assert out["n_unconverged"] == 0 and out["n_gsl_errors"] == 0
    nbytes, host_s = co.ingest_stats()
    assert nbytes == 2 * 2 * E * St * wire
    # given parameters from host vectors
    rng = np.random.default_rng(8)
    phi = rng.uniform(0.002, 0.02, St); p = rng.uniform(0.08, 0.2, St)
    out = co.run_host(th, rh, layout, phi=phi, expected=p, want_path=True)
    b = edlib.Batch(plan, St)
    b.run(test, ref, phi, p)
    assert out["calls"].tobytes() == b.calls().tobytes()
    assert np.array_equal(out["path"] if layout == 0 else out["path"].T, b.path())
    b.close()
    co.close()
    del keep


@pytest.mark.parametrize("kind", ["int32 pageable", "uint16 pageable", "int32 pinned", "uint16 pinned", "int32 pageable, narrowing off"])
def test_host_fed_slabs_stay_16_bits_wide_in_the_sample_major_table_mode(cohort_data, kind):
    """What the R-level entry runs (shim/edcore_shim.c: emit mode 2, R's column-major matrices, wire = 4): int32 blocks in pageable memory are
    narrowed to uint16 by the host threads that stage them and the slab stays uint16 on the device; uint16 blocks go up as they lie; int32 in
    pinned memory is read in place and stays int32.  The results are the device-resident int32 run's, bit for bit, either way -- and a slab with
    ONE count beyond 65 535 in its middle goes up 32 bits wide (the others still 16)."""
    edlib, plan, slabs, want, S = cohort_data
    if "want_mode_2" not in _CACHE:
        _CACHE["want_mode_2"] = _reference_results(edlib, plan, slabs, 2)
    want = _CACHE["want_mode_2"]
    test = np.concatenate([t for t, _ in slabs], axis=1)
    ref = np.concatenate([r for _, r in slabs], axis=1)
    St = test.shape[1]
    dt = np.uint16 if kind.startswith("uint16") else np.int32
    th, rh = np.ascontiguousarray(test.T.astype(dt)), np.ascontiguousarray(ref.T.astype(dt))
    keep = []
    if "pinned" in kind:
        pt, pr = edlib.PinnedArray(th.shape, dt), edlib.PinnedArray(rh.shape, dt)
        pt.array[...] = th; pr.array[...] = rh
        th, rh = pt.array, pr.array
        keep = [pt, pr]
    opts = dict(emit_mode=2, counts_layout=1)
    if kind.endswith("off"):
        opts["host_narrow"] = 0
    co = edlib.Cohort(plan, S, 2, **opts)
    calls = np.concatenate([w["calls"] for w in want])
    calls["sample"] += np.concatenate([np.full(len(w["calls"]), k * S, dtype=np.int32) for k, w in enumerate(want)])
    path = np.concatenate([w["path"] for w in want], axis=1)
    for rep in range(2):
        out = co.run_host(th, rh, 1, want_path=True)
        assert out["calls"].tobytes() == calls.tobytes()
        assert np.array_equal(out["path"].T, path)
        # (fitted from [samples][exons] counts here, from [exons][samples] in `want`: the same optimum to the fit's tolerance, so the decoration agrees
        #  to that and not to the bit; with GIVEN parameters -- below -- everything is bit for bit)
        assert np.allclose(out["phi"], np.concatenate([w["phi"] for w in want]), rtol=1e-8, atol=0)
        winfo = np.concatenate([w["info"] for w in want])
        for f in winfo.dtype.names:
            a, bq = out["info"][f], winfo[f]
            assert np.array_equal(a, bq) if a.dtype.kind in "iu" else np.allclose(a, bq, rtol=1e-6, atol=0), f
        if rep == 0:
            first = {k: np.array(out[k], copy=True) for k in ("calls", "info", "path", "phi", "expected")}
        else:                                                   # the same run twice: bit for bit
            assert all(first[k].tobytes() == np.asarray(out[k]).tobytes() for k in first)
    nbytes, _ = co.ingest_stats()
    narrowed = kind in ("int32 pageable", "uint16 pageable", "uint16 pinned")
    assert nbytes == 2 * 2 * plan.n_exons * St * (2 if narrowed else 4)
    assert co.n_wide_slabs() == 0
    co.close()
    if kind == "int32 pageable":
        # given parameters, and a count that does not fit 16 bits in the middle of the second slab
        rng = np.random.default_rng(8)
        phi = rng.uniform(0.002, 0.02, St); p = rng.uniform(0.08, 0.2, St)
        big_t, big_r = test.copy(), ref.copy()
        big_r[plan.n_exons // 2, S + S // 2] = 70_000
        big_t[7, S + 3] = 65_536
        b = edlib.Batch(plan, St)
        b.set_emit_mode(2)
        b.run(big_t, big_r, phi, p)
        co = edlib.Cohort(plan, S, 2, **opts)
        out = co.run_host(np.ascontiguousarray(big_t.T), np.ascontiguousarray(big_r.T), 1, phi=phi, expected=p, want_path=True)
        assert out["calls"].tobytes() == b.calls().tobytes() and out["info"].tobytes() == b.call_info().tobytes()
        assert np.array_equal(out["path"].T, b.path())
        assert co.n_wide_slabs() == 1
        # ... and a negative count (invalid input, the strict arithmetic's business): wide as well, same result as the int32 path
        neg = test.copy(); neg[11, 5] = -3
        b2 = edlib.Batch(plan, St); b2.set_emit_mode(2); b2.run(neg, ref, phi, p)
        out = co.run_host(np.ascontiguousarray(neg.T), np.ascontiguousarray(ref.T), 1, phi=phi, expected=p, want_path=True)
        assert out["calls"].tobytes() == b2.calls().tobytes() and np.array_equal(out["path"].T, b2.path())
        assert co.n_wide_slabs() == 2
        b.close(); b2.close(); co.close()
    del keep


def test_stage_ms_reports_the_last_fit_and_run_after_folding(cohort_data):
    """after the canonical fit -> run sequence ed_batch_stage_ms must report both, also after ed_batch_stage_ms_total"""
    edlib, plan, slabs, want, S = cohort_data
    test, ref = slabs[0]
    b = edlib.Batch(plan, S)
    b.enable_timing(True)
    dphi = edlib.DeviceArray(np.zeros(S)); dexp = edlib.DeviceArray(np.zeros(S))
    dt, dr = edlib.DeviceArray(test), edlib.DeviceArray(ref)
    for _ in range(2):
        b.fit(dt, dr, dphi, dexp)
        b.run(dt, dr, dphi, dexp)
    ms = b.stage_ms()
    assert ms["fit"] > 0 and ms["emissions"] > 0
    tot, nr, nf = b.stage_ms_total()
    assert nr == 2 and nf == 2
    ms2 = b.stage_ms()
    assert ms2["fit"] == ms["fit"] and ms2["emissions"] == ms["emissions"]
    b.close()


def test_container_to_calls_end_to_end(edlib, tmp_path):
    """The count-matrix container (EDCOUNT1, exomedepth_amd/io.py -- what stands where getBamCounts' data.frame is,
    R/countBamInGranges.R:298-370) feeds the pipeline directly: the memory-mapped count block is staged slab by slab through the
    pinned double buffer (no copy of the file in host memory), the reference sets and aggregate references are made on the device,
    and the calls equal the batch interface's on the same data."""
    from exomedepth_amd import io as edio, synth
    rng = np.random.default_rng(17)
    E, S = 9000, 96
    chrom_off, start, end = synth.exon_design(E, 5, seed=4)
    chromosome = np.repeat([str(c) for c in (3, 1, "X", 2, 7)], np.diff(chrom_off))           # unsorted chromosome labels
    lam = rng.lognormal(np.log(80.0), 0.7, E)
    counts = rng.poisson(lam[:, None] * rng.lognormal(0, 0.2, S)[None, :] * np.exp(rng.normal(0, 0.08, (E, S)))).astype(np.int32)
    path = str(tmp_path / "cohort.edcount")
    order = edio.write_counts(path, chromosome, start, end, counts)
    d = edio.read_counts(path, mmap=True)
    assert d["counts"].shape == (E, S) and np.array_equal(np.asarray(d["counts"]), counts[order])
    plan = edlib.Plan(d["chrom_off"], d["start"], d["end"])
    rs = edlib.cohort_select_reference_sets(np.asarray(d["counts"]), (d["end"] - d["start"]) / 1000.0, 3000)
    ref_h = rs["reference"].to_host().reshape(E, S)
    co = edlib.Cohort(plan, 40, 2)                                                             # slabs of 40 + 40 + 16 columns of the mapped file
    out = co.run_host(d["counts"], ref_h, 0, want_path=True)
    nbytes, _ = co.ingest_stats()
    assert nbytes == 2 * E * S * 4
    b = edlib.Batch(plan, S)
    b.run(np.asarray(d["counts"]), ref_h, out["phi"], out["expected"])
    assert out["calls"].tobytes() == b.calls().tobytes() and np.array_equal(out["path"], b.path())
    assert len(out["calls"]) > 0
    b.close(); co.close(); plan.close()


def _bins_reference(edlib, plan, slabs, B):
    out = []
    for test, ref in slabs:
        n = test.shape[1]
        b = edlib.Batch(plan, n)
        d = [edlib.DeviceArray(np.zeros((B, n))), edlib.DeviceArray(np.zeros((B + 1, n))), edlib.DeviceArray(np.zeros(n))]
        b.fit_bins(test, ref, B, *d)
        b.run_bins(test, ref, B, *d)
        out.append({"calls": b.calls().copy(), "info": b.call_info().copy(), "path": b.path().copy(), "loglik": b.loglik().copy(),
                    "phi_bins": d[0].to_host(), "edges": d[1].to_host(), "expected": d[2].to_host(), "form": b.fit_bins_form})
        b.close()
    return out


def _bins_check(edlib, plan, slab, B, got, want):
    """the fit at tolerance (the single-dispersion start of a slab that follows a deeper one sums its histograms in another geometry --
    an order of summation, DESIGN.md 4.5 -- so the last bits of the estimates depend on the batch object's history), complete.bins bit for
    bit, and everything downstream bit for bit GIVEN the pipeline's own parameters"""
    test, ref = slab
    n = test.shape[1]
    assert got["edges"].tobytes() == want["edges"].tobytes()
    assert np.max(np.abs(got["phi_bins"] - want["phi_bins"]) / want["phi_bins"]) < 1e-9
    assert np.max(np.abs(got["expected"] - want["expected"]) / want["expected"]) < 1e-10
    b = edlib.Batch(plan, n)
    b.run_bins(test, ref, B, got["phi_bins"], got["edges"], got["expected"])
    assert got["calls"].tobytes() == b.calls().tobytes()
    assert got["info"].tobytes() == b.call_info().tobytes()
    assert got["path"].tobytes() == b.path().tobytes()
    assert got["loglik"].tobytes() == b.loglik().tobytes()
    b.close()


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_depth_binned_model_through_the_cohort(cohort_data, in_flight):
    """option phi_bins = 3 (the reference's phi.bins, R/class_definition.R:120-147): every slab's grouped fit is issued without a host
    look at its outcome and settled at the first wait; results are those of ed_batch_fit_bins + ed_batch_run_bins slab by slab.  The third
    slab's reference counts are scaled beyond the histogram form's bins: that slab is declined on the device and done again per cell."""
    edlib, plan, slabs, want, S = cohort_data
    B = 3
    slabs = [(t, r) for t, r in slabs]
    slabs[2] = (slabs[2][0], slabs[2][1] * 60)
    ref_res = _bins_reference(edlib, plan, slabs, B)
    assert [w["form"] for w in ref_res] == [1, 1, 0, 1, 1]
    co = edlib.Cohort(plan, S, in_flight, phi_bins=B)
    dev = [(edlib.DeviceArray(t), edlib.DeviceArray(r)) for t, r in slabs]
    tickets = []
    for rounds in range(2):
        for i, (dt, dr) in enumerate(dev):
            n = slabs[i][0].shape[1]
            if len(tickets) >= in_flight:
                j = len(tickets) - in_flight
                got = co.results(tickets[j], slabs[j % len(slabs)][0].shape[1], path=True, loglik=True)
                _bins_check(edlib, plan, slabs[j % len(slabs)], B, got, ref_res[j % len(slabs)])
            tickets.append(co.submit(dt, dr, n_samples=n))
    for j in range(len(tickets) - in_flight, len(tickets)):
        got = co.results(tickets[j], slabs[j % len(slabs)][0].shape[1], path=True, loglik=True)
        _bins_check(edlib, plan, slabs[j % len(slabs)], B, got, ref_res[j % len(slabs)])
    with pytest.raises(edlib.EdError):                             # per-sample parameters cannot be given in this mode
        co.submit(dev[0][0], dev[0][1], phi=edlib.DeviceArray(np.full(S, 0.01)), expected=edlib.DeviceArray(np.full(S, 0.1)))
    co.close()
    # the whole cohort from host memory (R's layout)
    test = np.concatenate([t for t, _ in slabs], axis=1); ref = np.concatenate([r for _, r in slabs], axis=1)
    co = edlib.Cohort(plan, S, 2, phi_bins=B)
    out = co.run_host(np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T), 1, want_path=True)
    co.close()
    wp = np.concatenate([w["phi_bins"] for w in ref_res], axis=1)
    assert np.max(np.abs(out["phi_bins"] - wp) / wp) < 1e-9
    assert np.array_equal(out["edges"], np.concatenate([w["edges"] for w in ref_res], axis=1))
    b = edlib.Batch(plan, test.shape[1])
    b.run_bins(test, ref, B, out["phi_bins"], out["edges"], out["expected"])
    assert np.array_equal(out["path"].T, b.path()) and out["calls"].tobytes() == b.calls().tobytes()
    b.close()


def test_an_empty_depth_level_is_the_error_of_its_ticket(cohort_data):
    """R/class_definition.R:130-133 through the pipeline: the slab with a constant reference column fails when it is waited for; the
    slabs around it are unaffected"""
    edlib, plan, slabs, want, S = cohort_data
    B = 3
    bad = slabs[1][1].copy(); bad[:, 7] = 500
    ref_res = _bins_reference(edlib, plan, [slabs[0], slabs[3]], B)
    co = edlib.Cohort(plan, S, 2, phi_bins=B)
    d0 = (edlib.DeviceArray(slabs[0][0]), edlib.DeviceArray(slabs[0][1]))
    d1 = (edlib.DeviceArray(slabs[1][0]), edlib.DeviceArray(bad))
    d3 = (edlib.DeviceArray(slabs[3][0]), edlib.DeviceArray(slabs[3][1]))
    t0 = co.submit(*d0); t1 = co.submit(*d1)
    _bins_check(edlib, plan, slabs[0], B, co.results(t0, S, path=True, loglik=True), ref_res[0])
    t2 = co.submit(*d3)
    with pytest.raises(edlib.EdError, match="Binning did not happen properly"):
        co.wait(t1)
    _bins_check(edlib, plan, slabs[3], B, co.results(t2, S, path=True, loglik=True), ref_res[1])
    co.close()


def test_test_counts_from_the_host_references_on_the_device(cohort_data):
    """ed_cohort_submit_host_test: only the test matrix crosses the link (16-bit wire format, R's layout), the references are device-resident"""
    edlib, plan, slabs, want, S = cohort_data
    co = edlib.Cohort(plan, S, 2)
    tickets = []
    keep = []
    for i, (t, r) in enumerate(slabs):
        dr = edlib.DeviceArray(r); keep.append(dr)
        if i % 2 == 0:
            tickets.append(co.submit_host_test(np.ascontiguousarray(t.T.astype(np.uint16)), dr, 1))
        else:
            tickets.append(co.submit_host_test(t.astype(np.int32), dr, 0))
        if i >= 1:
            _same(co.results(tickets[i - 1], slabs[i - 1][0].shape[1], path=True), want[i - 1], ("calls", "info", "path", "phi", "expected"))
    _same(co.results(tickets[-1], slabs[-1][0].shape[1], path=True), want[-1], ("calls", "info", "path", "phi", "expected"))
    co.close()
