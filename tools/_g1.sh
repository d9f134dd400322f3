#!/bin/bash
python tools/fit_timing.py 2>&1 | grep depth
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_tables.py -x -q 2>&1 | tail -3
export TMPDIR=/tmp; mkdir -p gpurun_out/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o ft -- python tools/fit_timing.py 1 > gpurun_out/ks/ftlog 2>&1
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/ks/ft_kernel_stats.csv')))
for r in rows[1:]:
    if '(anonymous namespace)::' in r[0]:
        name = r[0].replace('(anonymous namespace)::', '').split('(')[0]
        print("%-28s calls %4s avg_ms %8.4f min %8.4f max %8.4f" % (name, r[1], float(r[3]) / 1e6, float(r[5])/1e6, float(r[6])/1e6))
PY
