"""Host-side logic of the interface mirror (no GPU): exon ordering of CallCNVs, R-style coercions,
synthetic generators, sample sharding."""
import numpy as np
import pytest

import exomedepth_amd as ed
from exomedepth_amd import api, dist, synth


def test_chromosome_order_follows_callcnvs():
    """reference R/class_definition.R:323-336: levels = '1'..'22' (those present) then the other names in
    first-seen order; exons ordered by (level, midpoint), ties in input order."""
    chrom = ["X", "2", "10", "2", "chrM", "1", "X", "10", "2"]
    start = [50, 300, 10, 100, 5, 70, 10, 10, 100]
    end = [60, 400, 20, 200, 9, 90, 20, 20, 200]
    order, levels, codes, off = ed.chromosome_order(chrom, start, end)
    assert levels == ["1", "2", "10", "X", "chrM"]
    assert [chrom[i] for i in order] == ["1", "2", "2", "2", "10", "10", "X", "X", "chrM"]
    assert order.tolist() == [5, 3, 8, 1, 2, 7, 6, 0, 4]      # the two identical chr2 exons keep input order
    assert off.tolist() == [0, 1, 4, 6, 8, 9]


def test_r_style_coercions():
    assert api._as_r_integer(np.array([1.9, -1.9, 3.0])).tolist() == [1, -1, 3]     # as.integer truncates
    assert api._signif(123456.0, 3) == 123000.0 and api._signif(0.0123456, 3) == 0.0123
    assert api._signif(-2.3456, 3) == -2.35 and api._signif(0.0, 3) == 0.0


def test_synthetic_design_and_counts_are_deterministic_and_sane():
    off, start, end = synth.exon_design(5000, 24, seed=3)
    off2, start2, end2 = synth.exon_design(5000, 24, seed=3)
    assert np.array_equal(start, start2) and off[-1] == 5000 and len(off) == 25
    for c in range(24):
        s = start[off[c]:off[c + 1]]
        assert np.all(np.diff(s) > 0)
    assert np.all(end > start) and start.dtype == np.int32
    test, ref, p, phi, state = synth.counts_numpy(off, 8, seed=4, n_segments=4)
    assert test.shape == (5000, 8) and test.dtype == np.int32 and np.all(test >= 0) and np.all(ref >= 0)
    frac = test.sum(axis=0) / (test.sum(axis=0) + ref.sum(axis=0))
    assert np.all(np.abs(frac - p) < 0.01)
    assert 0 < (state != 0).mean() < 0.05


def test_shard_bounds_partition_the_sample_axis():
    for S, W in ((8192, 8), (10, 3), (5, 8), (1024, 1)):
        b = [dist.shard_bounds(S, r, W) for r in range(W)]
        assert b[0][0] == 0 and b[-1][1] == S
        assert all(b[i][1] == b[i + 1][0] for i in range(W - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_exomedepth_low_coverage_guard():
    """reference R/class_definition.R:94-97: fewer than 5 bins with more than 5 reads -> empty object."""
    x = ed.ExomeDepth(np.array([0, 1, 2, 9, 9, 9, 9]), np.array([5, 5, 5, 50, 50, 50, 50]))
    assert x.phi.size == 0 and x.likelihood.shape == (0, 3)
    assert x.CallCNVs(["1"] * 7, range(7), range(1, 8), list("abcdefg")).CNV_calls == []


def test_count_container_roundtrip(tmp_path):
    from exomedepth_amd import io
    rng = np.random.default_rng(8)
    chrom = ["2"] * 40 + ["X"] * 10 + ["1"] * 50
    start = np.concatenate([np.sort(rng.integers(1, 10**6, 40)), np.sort(rng.integers(1, 10**6, 10)), np.sort(rng.integers(1, 10**6, 50))[::-1]])
    end = start + 100
    counts = rng.integers(0, 500, (100, 7))
    names = ["e%d" % i for i in range(100)]
    path = str(tmp_path / "cohort.edc")
    order = io.write_counts(path, chrom, start, end, counts, sample_names=list("ABCDEFG"), exon_names=names)
    d = io.read_counts(path)
    assert d["chrom_names"] == ["1", "2", "X"] and d["chrom_off"].tolist() == [0, 50, 90, 100]
    assert np.array_equal(d["counts"], counts[order]) and d["counts"].dtype == np.int32
    assert np.array_equal(d["start"], start[order]) and d["sample_names"] == list("ABCDEFG")
    assert d["exon_names"] == [names[i] for i in order]
    assert np.all(np.diff(d["start"][:50]) >= 0)            # chromosome 1 was given in descending order
    assert d["counts"].offset % 4096 == 0                    # page-aligned block, mappable straight into a Batch
    assert np.array_equal(io.read_counts(path, mmap=False)["counts"], d["counts"])
    import pytest
    with pytest.raises(ValueError):
        io.write_counts(path, chrom, start, end, counts + 0.5)


def test_bin_thinning_follows_r_seq_semantics():
    """selected[seq(1, length(selected), length(selected) / n.bins.reduced)] (reference R/optimize_reference_set.R:86).
    R's seq.default computes from + (0:n) * by with n = as.integer((to - from) / by + 1e-10), clips with pmin(., to), and
    a fractional subscript truncates.  Accumulating the step instead (pos += by) picks different bins for about a quarter
    of the (length, n) pairs: length 50, n 24 gives position 25 instead of R's 26 at k = 12 (1-based)."""
    import ctypes as C
    from fractions import Fraction
    from exomedepth_amd import _lib
    L = _lib.lib()

    def lib_positions(length, nred):
        out = np.zeros(nred + 2, dtype=np.int64)
        n = C.c_int64(0)
        _lib.check(L.ed_refset_thin_positions(length, nred, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value]

    got = lib_positions(50, 24)
    assert got[12] + 1 == 26                   # R: seq(1, 50, 50/24)[13] = 26.000000000000004 -> 26
    assert got[0] == 0 and got[-1] <= 49 and np.all(np.diff(got) >= 1)
    n_differs_from_accumulation = 0
    rng = np.random.default_rng(0)
    for _ in range(400):
        length = int(rng.integers(5, 200000)); nred = int(rng.integers(1, length))
        by = np.float64(length) / np.float64(nred)
        n = int((np.float64(length) - 1.0) / by + 1e-10)
        want = (np.minimum(1.0 + np.arange(n + 1, dtype=np.float64) * by, float(length))).astype(np.int64) - 1
        got = lib_positions(length, nred)
        assert np.array_equal(got, want), (length, nred)
        # exact rational positions differ from the binary64 ones only where k * by rounds across an integer
        exact = [int(1 + Fraction(k * length, nred)) - 1 for k in range(min(n + 1, 50))]
        assert all(abs(int(a) - b) <= 1 for a, b in zip(got[:50], exact))
        acc, v = [], 1.0
        while v <= length + 1e-10:
            acc.append(int(v) - 1); v += by
        n_differs_from_accumulation += (len(acc) != len(got)) or bool(np.any(np.array(acc) != got))
    assert n_differs_from_accumulation > 20    # the old accumulation really was a different function
    from oracle import refset_oracle
    for length, nred in ((50, 24), (9000, 700), (123457, 10000)):
        assert np.array_equal(refset_oracle.r_seq_thin(length, nred), lib_positions(length, nred))


def test_host_slabs_are_checked_before_they_are_uploaded():
    """Cohort.submit_host reads raw memory: arrays that are not in the layout the C entry walks are copied into it, windows of wider
    matrices stay in place (exomedepth_amd/api.py:_host_slab)"""
    from exomedepth_amd.api import _host_slab
    wide = np.arange(60, dtype=np.int32).reshape(6, 10)
    win = wide[:, :4]                                    # first columns of a wider matrix: in place, row pitch 10
    assert _host_slab(win, 0) is win and win.strides[0] == 40
    f = np.asfortranarray(wide)
    g = _host_slab(f, 0)
    assert g is not f and g.flags.c_contiguous and np.array_equal(g, wide)
    assert _host_slab(wide[:, ::2], 0).flags.c_contiguous and _host_slab(wide[::-1], 0).strides[0] > 0
    t = wide.T                                           # (n, n_exons) view of R's matrix: dense only after a copy
    assert _host_slab(np.ascontiguousarray(t), 1).flags.c_contiguous and _host_slab(t, 1).flags.c_contiguous
    assert np.array_equal(_host_slab(t, 1), t)
    with pytest.raises(ValueError):
        _host_slab(wide, 2)
    with pytest.raises(ValueError):
        _host_slab(np.zeros(5, np.int32), 0)
