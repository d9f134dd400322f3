"""Start / end (ms) of every kernel of this library across the last two complete bench steps of a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python bench.py --steps 8 --warmup 3 --cpu-samples 0
    python tools/trace_steps.py /tmp/tl/t_kernel_trace.csv
(start = the dispatch packet is taken up, not the first wave: a kernel that waits for CUs shows as a long one)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last three k_emit_batch launches and everything between them
idx = [i for i, r in enumerate(rows) if "k_emit_batch" in r["Kernel_Name"] or "k_emit_tab" in r["Kernel_Name"]]
a, b = idx[-7], idx[-3]   # (two launches per slab with the cohort pipeline's split: two whole steps)
t0 = int(rows[a]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0 or s > int(rows[b]["End_Timestamp"]): continue
    import re
    name = re.search(r"(k_\w+)", r["Kernel_Name"]).group(1)
    if "hg" in r["Kernel_Name"]: name = "hg:" + name
    print("%-18s %8.3f -> %8.3f ms  (%.3f)  q=%s" % (name, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?")))
