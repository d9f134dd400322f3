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
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python bench.py --steps 5 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 $EXTRA > $OUT/bench_ks.log 2>&1
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
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- python bench.py --steps 2 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 $EXTRA > $OUT/pmc_$N.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_${N}_counter_collection.csv > $OUT/pmc_$N.csv
  rm -f $OUT/pmc_${N}_counter_collection.csv $OUT/pmc_${N}_kernel_trace.csv
done
python - "$OUT" "$TAG" $EXTRA <<'PY'
import json, sys, time
sys.path.insert(0, ".")
from exomedepth_amd import _build
line = json.load(open(sys.argv[1] + "/bench_line.json"))
json.dump({"tag": sys.argv[2], "csrc_sha16": _build.csrc_sha16(), "pmc_steps": 3, "kernel_stats_steps": 6,
           "workload": {"exons": 200000, "samples_per_gpu": 1024, "kernel": line["roofline"]["kernel"],
                        "emission_launches_per_run": line["roofline"]["launches_per_step"],
                        "bench_flags": "defaults (cohort pipeline of the library, two slabs in flight, fit on) " + " ".join(sys.argv[3:])},
           "taken": time.strftime("%Y-%m-%d %H:%M:%S"), "bench_args": "--steps 2 --warmup 1 (PMC passes); --steps 5 --warmup 1 (kernel trace)"},
          open(sys.argv[1] + "/meta.json", "w"), indent=1)
PY
rm -f $OUT/ks_kernel_trace.csv $OUT/ks_domain_stats.csv $OUT/*agent_info.csv
ls -la $OUT
