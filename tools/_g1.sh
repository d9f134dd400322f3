#!/bin/bash
Q="--steps 12 --warmup 3 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --config1-steps 0 --pipeline 0"
for rep in 1 2; do
for P in 1 0; do
  echo "== pack $P"
  ED_VIT_PACK=$P timeout 200 python bench.py $Q 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['ms_per_step'],3), d['stage_ms'])"
done
done
