"""Every dispatch (this library's kernels AND the runtime's fill / copy kernels) between the end of one slab's emission launch and the
start of the next one's, from a rocprofv3 kernel trace:  python tools/trace_boundary.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
em = [i for i, r in enumerate(rows) if "k_emit_batch" in r["Kernel_Name"]]
a, b = em[-5], em[-3]        # end of a slab's second launch .. the next slab's two launches
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n.split("::")[-1].split("(")[0] if "anonymous" in n else n[:48]
    print("%-48s %9.3f -> %9.3f us  (%.1f)  q=%s" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")))
