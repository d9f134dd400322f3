export TMPDIR=/tmp; mkdir -p gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o ks -- python bench.py --steps 5 --warmup 1 --cpu-samples 0 "$@" > gpurun_out/ks/log 2>&1
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/ks/ks_kernel_stats.csv')))
for r in rows[1:]:
    if 'k_' in r[0]: print(r[0][-40:], r[1], r[3], r[2])
PY
rm -f gpurun_out/ks/ks_kernel_trace.csv
