#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_tables.py tests/test_gpu_cohort.py tests/test_gpu_fit_concordance.py tests/test_shim.py -x -q 2>&1 | tail -5
Q="--steps 12 --warmup 3 --cpu-samples 0 --kernel-alone 0 --verify-columns 4 --fit-concordance 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --config1-steps 0"
for rep in 1 2; do
  timeout 200 python bench.py $Q 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['ms_per_step'],3), d['stage_ms'], d['verify'])"
done
