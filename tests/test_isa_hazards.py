"""The shipped gfx950 machine code is free of the hazard that explains round 1's one unreproduced emission mismatch:
a VALU instruction reading an SGPR less than two wait states after a VALU instruction wrote it (v_readlane_b32 reloading
a spilled half of a Horner coefficient right in front of ed_pmath.h's inline-asm v_fma_f64; LLVM pads its own
instructions but not inline asm).  tools/isa_hazard_scan.py takes the code object out of libedcore.so and checks every
VALU instruction that names an SGPR operand.  No GPU needed: llvm-objdump ships with ROCm."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard_scan as hz  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(hz.OBJDUMP), reason="llvm-objdump (ROCm) not found")

# what the compiler produced in front of round 1's Horner step (k_emit_batch, exp of the Stirling tail), and its padded forms
HAZARD = """
0000000000001000 <k_with_hazard>:
	s_mov_b32 s4, 0x11111111                                   // 000000001000: BE8400FF 11111111
	s_mov_b32 s5, 0x3f811111                                   // 000000001008: BE8500FF 3F811111
	v_fma_f64 v[0:1], v[0:1], v[4:5], s[4:5]                   // 000000001010: D1CC0000 00120900
	v_readlane_b32 s5, v95, 13                                 // 000000001018: D2890005 00011B5F
	s_mov_b32 s4, s50                                          // 000000001020: BE840032
	v_fma_f64 v[0:1], v[0:1], v[4:5], s[4:5]                   // 000000001024: D1CC0000 00120900
0000000000002000 <k_padded_by_the_compiler>:
	v_readlane_b32 s5, v95, 13                                 // 000000002000: D2890005 00011B5F
	s_mov_b32 s4, s50                                          // 000000002008: BE840032
	s_nop 0                                                    // 00000000200C: BF800000
	v_fma_f64 v[0:1], v[0:1], v[4:5], s[4:5]                   // 000000002010: D1CC0000 00120900
	v_cmp_gt_i64_e64 s[10:11], s[24:25], v[8:9]                // 000000002018: D0E4000A 00021018
	v_mad_u64_u32 v[10:11], s[10:11], s18, v8, 0               // 000000002020: D1E80A0A 02021012
0000000000003000 <k_fixed_form>:
	v_readlane_b32 s28, v95, 13                                // 000000003000: D289001C 00011B5F
	v_mov_b32_e32 v2, v3                                       // 000000003008: 7E040303
	s_mov_b32 s28, 0x55555555                                  // 00000000300C: BE9C00FF 55555555
	s_mov_b32 s29, 0x3fa55555                                  // 000000003014: BE9D00FF 3FA55555
	v_fma_f64 v[0:1], v[0:1], v[4:5], s[28:29]                 // 00000000301C: D1CC0000 00720900
"""


def test_scanner_recognises_the_hazard_and_its_cures():
    n_funcs, n_readers, hazards = hz.scan(HAZARD)
    assert n_funcs == 3 and n_readers == 6
    assert [(f, ws) for f, w, r, ws in hazards] == [("k_with_hazard", 1)]
    assert "v_readlane_b32 s5" in hazards[0][1] and "s[4:5]" in hazards[0][2]


def test_shipped_library_has_no_valu_sgpr_write_then_read_hazard():
    from exomedepth_amd import _build
    if not os.path.exists(_build.LIB):
        _build.build()
    n_funcs, n_readers, hazards = hz.scan(hz.disassemble(hz.extract_code_object(_build.LIB)))
    assert n_funcs > 50 and n_readers > 20000            # the whole library was looked at
    assert hazards == []


def test_horner_constants_are_written_by_salu_inside_the_asm():
    """ed_pm_fma_k: every v_fma_f64 that takes its addend from s[28:29] is directly preceded by the two s_mov_b32 that
    write s28 and s29 from immediates."""
    from exomedepth_amd import _build
    if not os.path.exists(_build.LIB):
        _build.build()
    lines = [l.split("//")[0].strip() for l in hz.disassemble(hz.extract_code_object(_build.LIB)).split("\n")]
    lines = [l for l in lines if l]
    n = n_other = 0
    for i, l in enumerate(lines):
        if l.startswith("v_fma_f64") and l.endswith("s[28:29]"):
            if lines[i - 1].startswith("s_mov_b32 s29, ") and lines[i - 2].startswith("s_mov_b32 s28, "):
                n += 1              # the asm's triple
                continue
            # the register allocator may hand s[28:29] to a scalar double of its own (k_tab_build's logarithm keeps a coefficient there):
            # compiler-generated, hazard-padded by LLVM and covered by the scan above -- but no VALU may have written the pair just before
            n_other += 1
            for back in (1, 2):
                w = lines[i - back]
                assert not (w.startswith("v_") and any(r in w.split(",")[0] for r in ("s28", "s29", "s[28:29]"))), lines[i - 3:i + 1]
    assert n > 5000 and n_other < n // 20, (n, n_other)
