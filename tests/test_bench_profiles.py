"""bench.py quotes counter-derived figures (roofline.traffic, roofline.valu) only from a rocprofv3 profile that was taken on
the current build of the kernels: profiles/<tag>_meta.json carries the fingerprint of exomedepth_amd/csrc at profiling time.
CPU test of that logic on the committed profiles."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_stale_profiles_are_refused_and_matching_ones_parse(tmp_path, monkeypatch):
    import bench
    from exomedepth_amd import _build
    here = _build.csrc_sha16()
    assert len(here) == 16
    meta, why = bench.matching_profile()
    if meta is None:
        assert "withheld" in why and here in why
        pytest.skip("no committed profile matches the kernel sources (%s): bench.py withholds traffic / valu until "
                    "tools/profile_round.sh is run again" % here)
    assert meta["csrc_sha16"] == here
    kernel = meta.get("workload", {}).get("kernel", "k_emit_batch")         # the emission kernel of the mode the profile was taken in
    fig = bench.pmc_figures(meta, kernel, 200_000.0 * 1024, 2.27e10 if kernel == "k_emit_batch" else 1.1e11)
    assert fig["profile"] == meta["tag"]
    assert 5e9 < fig["traffic_bytes_per_step"] < 3e10                       # 33 B/cell materialised + table gathers
    if kernel == "k_emit_batch":                                            # strict mode: GSL's arithmetic, VALU-bound
        assert 1000 < fig["valu_lane_instructions_per_cell"] < 1600
        assert 0.5 < fig["valu_busy"] <= 1.0 and 0.3 < fig["valu_frac_of_fp64_peak"] < 1.0
    else:                                                                   # table-driven modes: a memory-streaming kernel
        assert 60 < fig["valu_lane_instructions_per_cell"] < 400 and fig["valu_busy"] < 0.9
        if kernel == "k_emit_tab_sm":
            assert fig["fetch_size_factor"] == 1.0 and fig["traffic_bytes_per_step"] < 8e9
    # a profile stamped with another fingerprint is not used
    fake = tmp_path / "profiles"
    fake.mkdir()
    (fake / "r99_x_meta.json").write_text(json.dumps({"tag": "r99_x", "csrc_sha16": "0" * 16}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    meta2, why2 = bench.matching_profile()
    assert meta2 is None and "withheld" in why2
