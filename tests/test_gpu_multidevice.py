"""One process, several devices (ed_multi_*; csrc/edmulti.inc): the cohort's slabs dealt to the devices from one queue,
one host thread per device, call tables merged in column order.  The samples are independent on this path (reference
vignette/vignette.Rnw:390-431 loops over them; R/class_definition.R:82-191, :311-419 see one test vector each), so the merged result
must be the single-device result bit for bit.  The GPU box has one device: it is named several times (two / three pipelines on one
GPU), which exercises the threads, the shares and the merge; on a node `devices=None` takes every visible GPU."""
import numpy as np
import pytest

import exomedepth_amd as ed
from exomedepth_amd import synth

pytestmark = pytest.mark.gpu


def _case(E=8000, S=230, seed=77):
    chrom_off, start, end = synth.exon_design(E, 5, seed)
    test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=4, mean_depth=80.0)
    return chrom_off, start, end, test, ref, p, phi


def _same(a, b):
    assert set(a) - {"devices"} == set(b) - {"devices"}
    for k in a:
        if k in ("devices",):
            continue
        if isinstance(a[k], dict) or isinstance(a[k], int):
            assert a[k] == b[k], k
        else:
            assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), k


@pytest.mark.parametrize("opts,layout", [({"emit_mode": 2, "counts_layout": 1}, 1), ({}, 0), ({"emit_mode": 1}, 0), ({"phi_bins": 3}, 0)])
def test_several_pipelines_give_the_single_device_result(edlib, opts, layout):
    chrom_off, start, end, test, ref, p, phi = _case()
    E, S = test.shape
    slab = 48                                       # 5 slabs, the last one ragged (38 columns)
    t_in, r_in = (np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)) if layout else (test, ref)
    plan = ed.Plan(chrom_off, start, end)
    co = ed.Cohort(plan, slab, 2, **opts)
    want = co.run_host(t_in, r_in, layout, want_path=True)
    co.close(); plan.close()
    assert len(want["calls"]) > 100
    for devs in ([0], [0, 0], [0, 0, 0], None):
        m = ed.MultiDevice(chrom_off, start, end, slab, devices=devs, **opts)
        got = m.run_host(t_in, r_in, layout, want_path=True)
        again = m.run_host(t_in, r_in, layout, want_path=True)          # the object is reusable
        D = m.n_devices
        m.close()
        _same(want, got); _same(want, again)
        dv = got["devices"]
        assert len(dv) == D and sum(d["slabs"] for d in dv) == 5 and sum(d["columns"] for d in dv) == S
        assert all(d["seconds"] > 0 for d in dv)


def test_a_throttled_pipeline_takes_fewer_slabs(edlib):
    """the queue at work: two pipelines on the one GPU, many small slabs -- whatever the split, the merged table is the single-device one; and with a
    pipeline that can hold one slab against one that holds four, both still finish the cohort between them"""
    chrom_off, start, end, test, ref, p, phi = _case(E=4000, S=600, seed=80)
    slab = 20                                       # 30 slabs
    opts = {"emit_mode": 2, "counts_layout": 1}
    t_in, r_in = np.ascontiguousarray(test.T), np.ascontiguousarray(ref.T)
    plan = ed.Plan(chrom_off, start, end)
    co = ed.Cohort(plan, slab, 2, **opts)
    want = co.run_host(t_in, r_in, 1, want_path=True)
    co.close(); plan.close()
    for devs, inflight in (([0, 0], 1), ([0, 0, 0], 4), ([0] * 8, 2)):
        m = ed.MultiDevice(chrom_off, start, end, slab, devices=devs, slabs_in_flight=inflight, **opts)
        for rep in range(3):
            got = m.run_host(t_in, r_in, 1, want_path=True)
            _same(want, got)
            assert sum(d["slabs"] for d in got["devices"]) == 30 and sum(d["columns"] for d in got["devices"]) == 600
        m.close()


def test_given_parameters_and_uint16_wire(edlib):
    chrom_off, start, end, test, ref, p, phi = _case(E=5000, S=96, seed=78)
    t16, r16 = np.ascontiguousarray(test.T.astype(np.uint16)), np.ascontiguousarray(ref.T.astype(np.uint16))
    plan = ed.Plan(chrom_off, start, end)
    co = ed.Cohort(plan, 32, 2, emit_mode=2, counts_layout=1)
    want = co.run_host(t16, r16, 1, phi=phi, expected=p, want_path=True)
    co.close(); plan.close()
    m = ed.MultiDevice(chrom_off, start, end, 32, devices=[0, 0], emit_mode=2, counts_layout=1)
    got = m.run_host(t16, r16, 1, phi=phi, expected=p, want_path=True)
    m.close()
    _same(want, got)
    assert np.array_equal(got["phi"], phi)


def test_errors_name_the_device_and_more_devices_than_slabs(edlib):
    chrom_off, start, end, test, ref, p, phi = _case(E=3000, S=40, seed=79)
    bad = ed.device_count()                         # (always one past the last visible device, on an 8-GPU node too)
    with pytest.raises(ed.EdError, match="device %d of %d visible" % (bad, bad)):
        ed.MultiDevice(chrom_off, start, end, 40, devices=[0, bad])
    with pytest.raises(ed.EdError, match="device -1"):
        ed.MultiDevice(chrom_off, start, end, 40, devices=[-1])
    # one slab, three pipelines: two of them get nothing to do
    plan = ed.Plan(chrom_off, start, end)
    co = ed.Cohort(plan, 40, 2)
    want = co.run_host(test, ref, 0)
    co.close(); plan.close()
    m = ed.MultiDevice(chrom_off, start, end, 40, devices=[0, 0, 0])
    got = m.run_host(test, ref, 0)
    assert sum(1 for d in got["devices"] if d["slabs"] > 0) == 1
    _same(want, got)
    # a failing share (phi given for some, NaN-free but phi_bins > 1 refuses given parameters): the caller gets the device's message
    m.close()
    m = ed.MultiDevice(chrom_off, start, end, 20, devices=[0, 0], phi_bins=2)
    with pytest.raises(ed.EdError, match=r"device 0 \(columns"):
        m.run_host(test, ref, 0, phi=phi, expected=p)
    m.close()
