#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_tables.py -x -q 2>&1 | tail -5
Q="--steps 12 --warmup 3 --cpu-samples 0 --verify-columns 4 --fit-concordance 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --config1-steps 0"
for rep in 1 2; do
for V in 1 0; do
  ED_SMP=$V timeout 200 python bench.py $Q 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('smp=$V step', round(d['ms_per_step'],3), 'emit live', round(d['roofline']['kernel_ms_per_step'],3), 'alone', round(d['roofline']['kernel_ms_alone'],3), {k: d['verify'][k] for k in ('loglik_beyond_1e-10','discordant_states','discordant_calls')})"
done
done
ED_SMP=1 bash tools/kernel_stats.sh 2>&1 | grep -v "void " | head -4; rm -rf gpurun_out/ks
