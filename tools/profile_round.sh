#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01_d
# Produces under gpurun_out/<tag>/: kernel stats of bench.py, and PMC summaries (separate passes, as the
# counters do not fit one pass and must not be mixed with other trace domains).
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
EXTRA="${@:2}"     # further bench.py flags (e.g. --emit-mode strict)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python bench.py --steps 5 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --regimes 0 --dropin 0 --regimes 0 --dropin 0 $EXTRA > $OUT/bench_ks.log 2>&1
grep -h '^{' $OUT/bench_ks.log > $OUT/bench_line.json
python - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.reader(open(out + "/ks_kernel_stats.csv")))
with open(out + "/kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(rows[0])
    for r in rows[1:]:
        if "(anonymous namespace)::k_" in r[0] or "(anonymous namespace)::hg" in r[0]: w.writerow(r)
PY
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$(echo $C | cut -d' ' -f1)
  [ "$N" = "SQ_WAVES" ] && N=SQ
  [ "$N" = "TCC_HIT_sum" ] && N=TCC
  [ "$N" = "SQ_LDS_BANK_CONFLICT" ] && N=LDS
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- python bench.py --steps 2 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --regimes 0 --dropin 0 --regimes 0 --dropin 0 $EXTRA > $OUT/pmc_$N.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_${N}_counter_collection.csv > $OUT/pmc_$N.csv
  rm -f $OUT/pmc_${N}_counter_collection.csv $OUT/pmc_${N}_kernel_trace.csv
done
# ---- the workflow leg (extra.workflow: upload -> reference sets for every sample -> calls): its kernels (k_rc_*, k_fit_accum_batched, ...)
#      in a kernel trace of their own and one PMC pass (MFMA work of k_rc_gram) ----
WF="--steps 1 --warmup 0 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --strict-steps 0 --regimes 0 --dropin 0 --workflow-reps 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o wf -- python bench.py $WF $EXTRA > $OUT/bench_wf.log 2>&1
grep -h '^{' $OUT/bench_wf.log > $OUT/bench_line_wf.json
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT -o pmc_WF -- python bench.py $WF $EXTRA > $OUT/pmc_WF.log 2>&1
python tools/pmc_summary.py $OUT/pmc_WF_counter_collection.csv > $OUT/pmc_WF.csv
rm -f $OUT/pmc_WF_counter_collection.csv $OUT/pmc_WF_kernel_trace.csv
python - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
rows = list(csv.reader(open(out + "/wf_kernel_stats.csv")))
keep = [r for r in rows[1:] if "(anonymous namespace)::k_" in r[0] or "(anonymous namespace)::hg" in r[0]]
with open(out + "/wf_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(rows[0]); w.writerows(keep)
# k_rc_gram: 2 * n4 * 32 * 32 flop per (block pair of the lower triangle incl. the diagonal, all slices together)
h = {k: i for i, k in enumerate(rows[0])}
line = json.load(open(out + "/bench_line_wf.json"))
wf = line["extra"]["workflow"]
S = int(line["config"]["samples_per_gpu"]); n = int(wf.get("n_bins_selected", wf.get("n_bins_reduced", 10000)))
Sp, n4 = (S + 31) // 32 * 32, (n + 3) // 4 * 4
nb = Sp // 32
flop = 2.0 * n4 * 1024 * (nb * (nb + 1) // 2)
for r in keep:
    if "k_rc_gram(" in r[0]:
        avg_ns = float(r[h["AverageNs"]])
        json.dump({"kernel": "k_rc_gram", "flop_per_launch": flop, "avg_ms": avg_ns / 1e6, "achieved_tflops_f64": flop / avg_ns / 1e3,
                   "peak_tflops_f64_matrix": 78.6, "frac": flop / avg_ns / 1e3 / 78.6, "n_rows": n4, "S_padded": Sp,
                   "note": "lower triangle of 32x32 blocks incl. the diagonal, v_mfma_f64_16x16x4_f64; peak = MI355X FP64 matrix 78.6 TF/s (MI355X_MICROARCH.md)"},
                  open(out + "/rc_gram.json", "w"), indent=1)
PY
rm -f $OUT/wf_kernel_trace.csv $OUT/wf_domain_stats.csv
python - "$OUT" "$TAG" $EXTRA <<'PY'
import csv, json, sys, time
sys.path.insert(0, ".")
from exomedepth_amd import _build
line = json.load(open(sys.argv[1] + "/bench_line.json"))
# The emission launches of the pipeline's lanes run SIDE BY SIDE (two lanes at the default four slabs in flight): a launch's own duration is then not
# the chip's time for it.  From the kernel trace of the timed steps: the union of the launches' intervals per launch (= chip time during which an
# emission launch is active, per launch) and how many are active on average while any is.
kernel = line["roofline"]["kernel"]
import re
pat = re.compile(r"::" + re.escape(kernel) + r"(<[^>]*>)?\(")          # (a template kernel's name carries its arguments)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1] + "/ks_kernel_trace.csv")) if pat.search(r["Kernel_Name"]))
n_lp = max(1, int(line["roofline"]["launches_per_step"]))
iv = iv[-int(line["steps"]) * n_lp:]                    # the timed steps' launches (the warm-up and priming runs come first)
union = 0; cur_s, cur_e = iv[0]
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union += cur_e - cur_s
# per kernel over the TIMED steps only (from the first timed emission launch on): launches, mean own duration, union of the intervals per launch --
# the set-up and warm-up launches (which run nearly alone) are not in these averages.  bench.py's roofline.frac = bytes per launch / union_ms_per_launch
# of the emission kernel / 8 TB/s; its kernel_ms_own = mean_ms.
t_begin = iv[0][0]
per = {}
for r in csv.DictReader(open(sys.argv[1] + "/ks_kernel_trace.csv")):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a < t_begin or "(anonymous namespace)::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    per.setdefault(name, []).append((a, b))
with open(sys.argv[1] + "/kernel_stats_timed.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "launches", "mean_ms", "union_ms_per_launch", "launches_in_flight_while_any", "total_ms"])
    for name, lst in sorted(per.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        lst.sort(); u = 0; cs, ce = lst[0]
        for a, b in lst[1:]:
            if a > ce: u += ce - cs; cs, ce = a, b
            else: ce = max(ce, b)
        u += ce - cs; tot = sum(b - a for a, b in lst)
        w.writerow([name, len(lst), "%.4f" % (tot / len(lst) / 1e6), "%.4f" % (u / len(lst) / 1e6), "%.3f" % (tot / u), "%.3f" % (tot / 1e6)])
overlap = {"launches": len(iv), "mean_ms_per_launch": sum(e_ - s_ for s_, e_ in iv) / len(iv) / 1e6, "union_ms_per_launch": union / len(iv) / 1e6,
           "launches_in_flight_while_any": sum(e_ - s_ for s_, e_ in iv) / union, "span_ms_per_launch": (iv[-1][1] - iv[0][0]) / len(iv) / 1e6}
json.dump({"tag": sys.argv[2], "csrc_sha16": _build.csrc_sha16(), "pmc_steps": 3, "kernel_stats_steps": 6, "emission_overlap": overlap,
           "workload": {"exons": 200000, "samples_per_gpu": 1024, "kernel": line["roofline"]["kernel"],
                        "emission_launches_per_run": line["roofline"]["launches_per_step"],
                        "bench_flags": "defaults (cohort pipeline of the library, %d slabs in flight, fit on) " % line["config"]["batches_in_flight"] + " ".join(sys.argv[3:])},
           "taken": time.strftime("%Y-%m-%d %H:%M:%S"), "bench_args": "--steps 2 --warmup 1 (PMC passes); --steps 5 --warmup 1 (kernel trace)"},
          open(sys.argv[1] + "/meta.json", "w"), indent=1)
PY
rm -f $OUT/ks_kernel_trace.csv $OUT/ks_domain_stats.csv $OUT/*agent_info.csv
ls -la $OUT
