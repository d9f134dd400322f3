#!/bin/bash
timeout 600 python bench.py --samples 8192 --steps 3 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 2 --fit-concordance 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --config1-steps 0 2>&1 | tail -3 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('S=8192 step', round(d['ms_per_step'],2), d['value'], d['verify'], d['n_calls'])
    else: print(l[:300])"
