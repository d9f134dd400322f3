export TMPDIR=/tmp; mkdir -p gpurun_out/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o ks -- python bench.py --steps 5 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --regimes 0 --dropin 0 --regimes 0 --dropin 0 "$@" > gpurun_out/ks/log 2>&1
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/ks/ks_kernel_stats.csv')))
for r in rows[1:]:
    if '(anonymous namespace)::' in r[0]:
        name = r[0].replace('(anonymous namespace)::', '').split('(')[0]
        print("%-28s calls %4s avg_ms %8.4f total_ms %9.3f" % (name, r[1], float(r[3]) / 1e6, float(r[2]) / 1e6))
PY
rm -f gpurun_out/ks/ks_kernel_trace.csv
