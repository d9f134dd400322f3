"""Copy the summaries of a profiling run from the scratch area into profiles/ (tracked):
    python tools/profile_publish.py r02_a          gpurun_out/r02_a/{kernel_stats.csv, pmc_*.csv, meta.json, bench_line.json}
                                                   -> profiles/r02_a_*.  bench.py quotes counter figures only from a profile
whose meta.json carries the fingerprint of the current kernel sources (exomedepth_amd/_build.py::csrc_sha16)."""
import glob, os, shutil, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
n = 0
for f in sorted(glob.glob(os.path.join(src, "*"))):
    b = os.path.basename(f)
    if b in ("kernel_stats.csv", "kernel_stats_timed.csv", "wf_kernel_stats.csv", "rc_gram.json", "meta.json", "bench_line.json", "bench_line_wf.json") or (b.startswith("pmc_") and b.endswith(".csv")) or b.startswith("bench_") and b.endswith(".json"):
        shutil.copy(f, os.path.join("profiles", "%s_%s" % (tag, b)))
        n += 1
print("published %d files as profiles/%s_*" % (n, tag))
