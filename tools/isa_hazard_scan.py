#!/usr/bin/env python3
"""Scan the SHIPPED gfx950 machine code of libedcore.so for the hazard behind round 1's one unexplained emission mismatch.

gfx940 / gfx950 need 2 wait states between a VALU instruction that WRITES an SGPR (v_readlane_b32, v_readfirstlane_b32,
v_cmp* / v_div_scale / carry-out forms with an SGPR destination) and a VALU instruction that READS that SGPR as an operand
(LLVM: GCNHazardRecognizer::checkVALUHazards, VALUWriteSGPRVALUReadWaitstates = 2).  The compiler pads its own
instructions with s_nop; it cannot see inside inline asm.  ed_pmath.h's Horner step is inline asm (v_fma_f64 with the
coefficient in an SGPR pair); when the coefficient was an "s" operand the register allocator reloaded spilled halves
with v_readlane_b32 right in front of it (53 such places in the build that preceded the fix, 3 of them in k_emit_batch).

This tool takes the code object out of the fat binary, disassembles it with llvm-objdump and checks EVERY VALU
instruction that names an SGPR operand -- compiler-generated or asm -- against the VALU SGPR writers less than two wait
states before it (straight-line predecessors; an instruction = 1 wait state, s_nop N = N + 1).

    python tools/isa_hazard_scan.py [exomedepth_amd/libedcore.so]      exit status 1 if a hazard is found
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = next((p for p in ("/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/llvm/bin/llvm-objdump") if os.path.exists(p)), "llvm-objdump")


def extract_code_object(so_path, arch="gfx950"):
    """the clang offload bundle inside the shared library -> bytes of the code object for `arch`"""
    data = open(so_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    i = data.find(magic)
    while i >= 0:
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if arch in triple and sz > 0:
                return data[i + o:i + o + sz]
        i = data.find(magic, i + 1)
    raise RuntimeError("no %s code object found in %s" % (arch, so_path))


def disassemble(code_object):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", f.name], check=True, capture_output=True, text=True).stdout


_sreg = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b|\bvcc\b")


def _sgprs(text):
    regs = set()
    for m in _sreg.finditer(text):
        if m.group(1) is not None:
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        elif m.group(3) is not None:
            regs.add(int(m.group(3)))
        else:
            regs.update((106, 107))          # vcc
    return regs


def scan(disassembly):
    """-> (functions, VALU instructions reading an SGPR, hazards [(function, writer, reader, wait states between)])"""
    hazards = []
    n_readers = 0
    n_funcs = 0
    func = None
    window = []          # (text, is_valu, sgprs written by a VALU, wait states it provides)
    for line in disassembly.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            func = m.group(1)
            n_funcs += 1
            window = []
            continue
        t = line.split("//")[0].strip()
        if not t or func is None or t.endswith(":"):
            continue
        parts = t.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        is_valu = op.startswith("v_")
        written = set()
        read = set()
        if is_valu and args:
            # destinations: the first operand; VOP3 compares / div_scale / carry forms have an SGPR destination first or second
            ndst = 2 if (op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32", "v_addc", "v_subb", "v_subbrev")) or "_co_" in op) else 1
            for a in args[:ndst]:
                written |= _sgprs(a)
            if op.startswith("v_cmp") and op.endswith("_e32"):
                written |= {106, 107}                      # implicit vcc
            for a in args[ndst:]:
                read |= _sgprs(a)
            if op.startswith(("v_readlane", "v_writelane")) and len(args) >= 3:
                read -= _sgprs(args[2])                    # the lane select has its own (4 wait state) rule, the compiler's business
        if read:
            n_readers += 1
            ws = 0
            for prev_t, prev_valu, prev_written, prev_ws in reversed(window):
                if ws >= 2:
                    break
                if prev_valu and (prev_written & read):
                    hazards.append((func, prev_t, t, ws))
                ws += prev_ws
        nop = (int(args[0], 0) + 1) if (op == "s_nop" and args) else 1
        window.append((t, is_valu, written if is_valu else set(), nop))
        if len(window) > 8:
            window.pop(0)
    return n_funcs, n_readers, hazards


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "exomedepth_amd", "libedcore.so")
    n_funcs, n_readers, hazards = scan(disassemble(extract_code_object(so)))
    by_func = {}
    for f, w, r, ws in hazards:
        by_func[f] = by_func.get(f, 0) + 1
    print(json.dumps({"library": os.path.basename(so), "functions": n_funcs, "valu_instructions_reading_an_sgpr": n_readers,
                      "hazards": len(hazards), "by_function": by_func,
                      "examples": [{"function": f, "writer": w, "reader": r, "wait_states_between": ws} for f, w, r, ws in hazards[:6]]}, indent=1))
    return 1 if hazards else 0


if __name__ == "__main__":
    sys.exit(main())
