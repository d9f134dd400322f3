#!/bin/bash
# One PMC pass (SQ_INSTS_VALU & co.) over a short bench run: VALU lane-instructions per cell of k_emit_batch.
#   tools/pmc_quick.sh <tag>        -> gpurun_out/<tag>/pmc_quick.csv
set -u
TAG=${1:-pmcq}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT -o pq -- \
  python bench.py --steps 2 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 > $OUT/pq.log 2>&1
python tools/pmc_summary.py $OUT/pq_counter_collection.csv > $OUT/pmc_quick.csv
rm -f $OUT/pq_counter_collection.csv $OUT/pq_kernel_trace.csv $OUT/*agent_info.csv
python - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/pmc_quick.csv")))
tot = {}
for r in rows:
    if r["kernel"] == "k_emit_batch":
        tot[r["counter"]] = float(r["mean_per_launch"]) * int(r["launches"])
        n = int(r["launches"])
runs = [int(r["launches"]) for r in rows if r["kernel"] == "k_sample_consts"][0]
cells = 200000 * 1024
print("k_emit_batch: launches %d, runs %d" % (n, runs))
print("VALU lane-instructions per cell: %.1f" % (tot["SQ_INSTS_VALU"] * 64 / runs / cells))
print("SALU/VALU %.3f  LDS insts per cell %.1f" % (tot["SQ_INSTS_SALU"] / tot["SQ_INSTS_VALU"], tot["SQ_INSTS_LDS"] * 64 / runs / cells))
PY
