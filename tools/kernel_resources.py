"""Registers, LDS, scratch and occupancy of every kernel of libedcore.so as the compiler reports them (hipcc -Rpass-analysis=kernel-resource-usage;
needs no GPU).  Which kernels can share a CU -- and which exclude each other -- follows from these numbers: a SIMD has 512 VGPRs, a CU 160 KB of LDS.
    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exomedepth_amd import _build

out = os.path.join(tempfile.mkdtemp(), "res.so")
cmd = [_build.hipcc()] + _build.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", out] + [os.path.join(_build.CSRC, s) for s in _build.SOURCES]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stderr[-2000:])
rows = []
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    g = lambda k: (re.search(k + r": (\S+)", b) or [None, "?"])[1]
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g(r"ScratchSize \[bytes/lane\]")))
print("kernel sources %s (exomedepth_amd._build.csrc_sha16)" % _build.csrc_sha16())
print("%-34s %5s %5s %5s %10s %9s %8s" % ("kernel", "VGPR", "AGPR", "SGPR", "waves/SIMD", "LDS B/WG", "scratch"))
for row in sorted(set(rows)):
    print("%-34s %5s %5s %5s %10s %9s %8s" % row)
