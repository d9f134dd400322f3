import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/libed_oracle.so), built on demand with gcc."""
    from oracle import edoracle
    edoracle.build()
    return edoracle


@pytest.fixture(scope="session")
def edlib():
    """The product library; GPU tests fail loudly if it is missing or no device is usable."""
    # When PyTorch shares the process it must bring up its HIP runtime first, so that libedcore.so binds to
    # the same libamdhip64 instead of loading a second copy (two runtimes in one process do not both see
    # the device).  bench.py imports torch first for the same reason.
    try:
        if os.environ.get("ED_LIB_VARIANT") in ("asan", "tsan"):    # tools/sanitize.sh: torch's HIP start-up does not survive the preloaded runtime
            raise ImportError
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import exomedepth_amd
    from exomedepth_amd import _build, _lib
    if not os.path.exists(_build.LIB):   # a tree that was never built (hipcc is on the GPU box too); no fallback exists
        _build.build()
    L = _lib.lib()
    assert L.ed_device_count() > 0, "no HIP device visible: GPU tests must run on the MI355X box"
    return exomedepth_amd
