"""The C-ABI library loads and exports every symbol include/exomedepth_amd.h declares; without a GPU
every compute entry fails loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "exomedepth_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ed_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from exomedepth_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)
    assert b"gfx950" in _lib.lib().ed_version()


def test_no_silent_cpu_fallback():
    from exomedepth_amd import EdError, _lib
    import exomedepth_amd as ed
    if _lib.lib().ed_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(EdError, match="no usable HIP device"):
        ed.get_loglike_matrix(0.01, 0.2, np.array([10], np.int32), np.array([2], np.int32))
    with pytest.raises(EdError, match="no usable HIP device"):
        ed.Plan([0, 2], [1, 100], [50, 150])
    with pytest.raises(EdError):
        ed.viterbi_hmm(np.eye(3), np.zeros((4, 3)), np.arange(4), 1.0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under exomedepth_amd/ may import, link or load it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "exomedepth_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                t = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle|libed_oracle|edoracle|#include\s+[\"<][^\n]*oracle", t, flags=re.M):
                    bad.append(f)
    assert not bad, bad
