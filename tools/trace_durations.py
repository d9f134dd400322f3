"""Print per-launch durations (ms) of this library's kernels from a rocprofv3 kernel_trace csv."""
import csv, sys
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
for r in csv.DictReader(open(sys.argv[1])):
    if pat in r["Kernel_Name"] and "anonymous" in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("::")[1].split("(")[0]
        print("%-18s %9.3f ms  vgpr=%s sgpr=%s scratch=%s lds=%s" % (
            name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("VGPR_Count"), r.get("SGPR_Count"),
            r.get("Scratch_Size"), r.get("LDS_Block_Size")))
