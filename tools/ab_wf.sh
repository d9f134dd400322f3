#!/bin/bash
# A/B of the workflow leg on one box: the shipped library against a variant; prints reference_sets_ms, calls_ms, total_ms.   tools/ab_wf.sh <variant> [rounds]
V=${1:-old}; N=${2:-2}
for i in $(seq 1 $N); do
  for v in "" "$V"; do
    f=""; [ -n "$v" ] && f="--lib-variant $v"
    python bench.py --cpu-samples 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --strict-steps 0 --kernel-alone 0 --steps 3 --warmup 1 --workflow-reps 3 $f 2>/dev/null | \
      python -c "import json,sys; w=json.loads(sys.stdin.read())['extra']['workflow']; print('${v:-new}', round(w['reference_sets_ms'],2), round(w['calls_ms'],2), round(w['total_ms'],2), round(w['back_to_back']['ms_per_cohort'],2))"
  done
done
