"""BASELINE.json configs[0] through the product: the reference's bundled data (fixture
tests/golden/exomecount_chr1.npz, from data/ExomeCount.RData) run with the mirror of the reference's own
workflow -- new('ExomeDepth', test, reference) then CallCNVs(...) (vignette/vignette.Rnw:191-252) -- on the GPU."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_bundled_exomecount_end_to_end(edlib, oracle):
    d = np.load(os.path.join(G, "exomecount_chr1.npz"))
    exp = np.load(os.path.join(G, "config1_expected.npz"))
    summ = json.load(open(os.path.join(G, "config1_summary.json")))
    start, end, counts = d["start"], d["end"], d["counts"].astype(np.float64)
    n = start.size
    chrom = ["chr1"] * n
    names = ["exon%d" % i for i in range(n)]
    for i in range(4):
        test = counts[:, i]
        ref = counts.sum(axis=1) - counts[:, i]
        x = edlib.ExomeDepth(test, ref)
        # fitted phi / expected vs the checker's long-double MLE (stored in the fixture)
        assert abs(x.phi[0] - float(exp["phi%d" % i])) / float(exp["phi%d" % i]) < 1e-8
        assert abs(x.expected[0] - float(exp["p%d" % i])) / float(exp["p%d" % i]) < 1e-8
        # likelihood given the device's own (phi, p): bit-identical to the checker's portable flavour
        ell, _ = oracle.get_loglike_matrix(x.phi[0], x.expected[0], (test + ref).astype(np.int32), test.astype(np.int32),
                                           1.0, oracle.PORTABLE)
        assert np.array_equal(np.ascontiguousarray(x.likelihood).view(np.int64), np.ascontiguousarray(ell).view(np.int64))
        x.CallCNVs(chrom, start, end, names)
        epath, ecalls = oracle.callcnvs(ell, np.array([0, n], np.int32), start, end)
        assert np.array_equal(x.Viterbi_path.astype(np.int8), epath)
        got = np.array([[c["start.p"], c["end.p"], {"deletion": 1, "duplication": 2}[c["type"]], c["nexons"]] for c in x.CNV_calls])
        assert np.array_equal(got, ecalls.astype(np.int64))
        # ... and equal to the stored result of the reference's arithmetic (libm flavour, MLE parameters)
        assert np.array_equal(epath, exp["path%d" % i]) and np.array_equal(ecalls, exp["calls%d" % i])
        assert len(x.CNV_calls) == summ["sample%d" % (i + 1)]["ncalls"]
        # decoration (R/class_definition.R:379-405)
        for c in x.CNV_calls[:5]:
            s, e = c["start.p"] - 1, c["end.p"] - 1
            assert c["id"] == "chr1:%d-%d" % (start[s], end[e]) and c["chromosome"] == "chr1"
            assert c["reads.observed"] == test[s:e + 1].sum()
            assert c["reads.expected"] == int(np.sum((test + ref)[s:e + 1] * x.expected[s:e + 1]))
            assert (c["BF"] > 0) and (c["type"] in ("deletion", "duplication"))
        assert abs(x.cor_test_reference - np.corrcoef(test, ref)[0, 1]) < 1e-12
        # TestCNV (R/class_definition.R:243-256): positive control on the first call, negative on a quiet window
        c0 = x.CNV_calls[0]
        assert x.TestCNV("chr1", c0["start"], c0["end"], c0["type"]) > 0
        assert x.TestCNV("chr1", int(start[100]), int(end[140]), "deletion") < 0
    if i == 3:
        assert len(x.CNV_calls) > 0
