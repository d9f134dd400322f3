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


def test_no_exception_can_leave_an_entry_point():
    """The callers are C (R's .Call, ctypes): every int-returning export of the library is a function-try-block closed by ED_CATCH
    (csrc/edcore.hip::ed_caught turns what was thrown into an error code + ed_last_error()).  One-line accessors hold nothing that throws."""
    csrc = os.path.join(ROOT, "exomedepth_amd", "csrc")
    n = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".inc")):
            continue
        lines = open(os.path.join(csrc, f)).read().split("\n")
        for i, l in enumerate(lines):
            if not l.startswith("ED_EXPORT int "):
                continue
            if l.rstrip().endswith("}") and "{" in l:
                assert "std::" not in l and "new " not in l, (f, i + 1)
                continue
            j = i
            while not lines[j].startswith(("{", "try {")):
                j += 1
            name = re.match(r"ED_EXPORT int (\w+)\(", l).group(1)
            assert lines[j] == "try {", "%s:%d %s is not a function-try-block" % (f, i + 1, name)
            k = j + 1
            while lines[k] != "}":
                k += 1
            assert lines[k + 1] == 'ED_CATCH("%s")' % name, "%s:%d %s" % (f, k + 2, name)
            n += 1
    assert n >= 90
