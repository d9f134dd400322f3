"""Timeline of the library's kernels over the last `n` steps of a rocprofv3 kernel trace of bench.py (a step = one emission launch of
the headline mode), plus per-kernel averages over those steps and the busy fraction of the interval:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python bench.py --steps 8 --warmup 3 --kernel-alone 0 ...
    python tools/timeline.py /tmp/tl/.../t_kernel_trace.csv [n_steps] [emission kernel name] [steps left out at the end]"""
import csv, re, sys
from collections import defaultdict
ALL = len(sys.argv) > 5 and sys.argv[5] == "all"      # also the runtime's own kernels (fills, copies)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ALL or ("anonymous" in r["Kernel_Name"] and re.search(r"(k_\w+)", r["Kernel_Name"]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ek = sys.argv[3] if len(sys.argv) > 3 else "k_emit_tab_sm"
back = int(sys.argv[4]) if len(sys.argv) > 4 else 2       # steps left out at the end of the run (the last ones have no next slab to fit)
def nm(r):
    m = re.search(r"(k_\w+)", r["Kernel_Name"]) if "anonymous" in r["Kernel_Name"] else None
    if not m:
        return "[" + r["Kernel_Name"][:28] + "]"
    name = m.group(1)
    return ("hg:" if "::hg" in r["Kernel_Name"] else "") + name
idx = [i for i, r in enumerate(rows) if nm(r) == ek]
a, b = idx[-(n + 1) - back], idx[-1 - back]
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
print("interval: %d steps, %.3f ms per step (start of %s to start of %s)" % (n, (t1 - t0) / 1e6 / n, ek, ek))
agg = defaultdict(lambda: [0, 0.0])
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s >= t1:
        continue
    print("%-20s %8.3f -> %8.3f ms  (%.3f)  q=%s" % (nm(r), (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?")))
    agg[nm(r)][0] += 1; agg[nm(r)][1] += (e - s) / 1e6
    ev.append((s, 1)); ev.append((min(e, t1), -1))
ev.sort()
busy = 0; depth = 0; last = t0; over = defaultdict(float)
for t, d in ev:
    over[depth] += t - last
    last = t; depth += d
over[depth] += t1 - last
print("kernels in flight -> share of the interval:", {k: round(v / (t1 - t0), 3) for k, v in sorted(over.items())})
print("per step: " + ", ".join("%s %.3f ms x %.1f" % (k, v[1] / v[0], v[0] / n) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])))
print("sum of kernel durations per step: %.3f ms" % (sum(v[1] for v in agg.values()) / n))
