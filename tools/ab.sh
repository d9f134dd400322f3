#!/bin/bash
# A/B on one box: alternate the shipped library and a variant (exomedepth_amd/libedcore_<name>.so); prints ms per step, the live
# emission time per step and the emission launches alone.   tools/ab.sh <variant> [rounds] [extra bench flags]
V=${1:-old}; N=${2:-3}; shift; shift
for i in $(seq 1 $N); do
  for v in "" "$V"; do
    f=""; [ -n "$v" ] && f="--lib-variant $v"
    python bench.py --cpu-samples 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --steps 20 $f "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${v:-new}', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_per_step'],3), round(d['roofline']['kernel_ms_alone'],3))"
  done
done
