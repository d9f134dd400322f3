"""Summarise a rocprofv3 counter_collection csv: per kernel of this library, mean counter value per launch."""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"\(anonymous namespace\)::((?:hg\d::)?k_\w+)", k)   # (search: a template kernel's name starts with its return type)   # hgN:: = a histogram geometry of the fit
    if not m:
        continue
    name = m.group(1)
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,launches,mean_per_launch")
for n in sorted(acc):
    for c in sorted(acc[n]):
        v = acc[n][c]
        print("%s,%s,%d,%.6g" % (n, c, len(v), sum(v) / len(v)))
