#!/bin/bash
Q="--steps 12 --warmup 3 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --config1-steps 0"
for rep in 1 2; do
for V in -1 1 0; do
  timeout 200 python bench.py $Q --tables-early $V 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tables_early=$V step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['stage_ms'].items()})"
done
done
