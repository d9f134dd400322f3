#!/bin/bash
export TMPDIR=/tmp; mkdir -p /tmp/wf
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/wf -o wf -- python bench.py --steps 1 --warmup 0 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --stage-inputs 0 --strict-steps 0 --config1-steps 0 --workflow-reps 2 > /tmp/wf/log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/wf/**/wf_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
acc = [d for n, d in seq if n == 'k_fit_accum']
print(len(acc), [round(x) for x in acc])
PY
