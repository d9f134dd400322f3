"""Print start/end (ms, relative to the first listed kernel) of this library's kernels in the LAST bench step of a
rocprofv3 kernel_trace csv:  python tools/trace_timeline.py <csv> [first_kernel_of_a_step]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous" in r["Kernel_Name"]]
first = sys.argv[2] if len(sys.argv) > 2 else "k_sample_consts"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
rows = rows[idx[-1]:] if idx else rows
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("::")[1].split("(")[0]
    print("%-16s %8.3f -> %8.3f ms  (%.3f)  queue=%s" % (name, (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
                                           (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?")))
