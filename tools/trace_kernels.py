"""Durations, in launch order, of the kernels whose name contains a pattern, from a rocprofv3 kernel trace (csv).
    python tools/trace_kernels.py trace.csv k_fit_accum [last N]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
for r in rows[-n:]:
    print("%.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6), end=" ")
print()
