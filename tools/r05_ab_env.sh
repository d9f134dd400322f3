#!/bin/bash
# A/B of environment settings on one box: tools/r05_ab_env.sh <rounds> "VAR=a" "VAR=b" ...   (bench default flags, headline only)
N=${1:-3}; shift
for i in $(seq 1 $N); do
  for f in "$@"; do
    env $f python bench.py --cpu-samples 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --regimes 0 --dropin 0 --steps 20 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('[%s]' % '$f', round(d['ms_per_step'],3), 'emit', round(d['roofline']['kernel_ms_per_step'],3), 'alone', round(d['roofline']['kernel_ms_alone'],3), {k: round(v,2) for k,v in s.items()})"
  done
done
