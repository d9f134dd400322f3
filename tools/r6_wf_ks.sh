# kernel stats of the workflow leg (reference sets + calls)
export TMPDIR=/tmp; mkdir -p gpurun_out/wfks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wfks -o wf -- python bench.py --steps 1 --warmup 0 --stage-inputs 0 --strict-steps 0 --config1-steps 0 --cpu-samples 0 --fit-concordance 0 --verify-columns 0 --kernel-alone 0 --regimes 0 --dropin 0 --workflow-reps 3 > gpurun_out/wfks/log 2>&1
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/wfks/wf_kernel_stats.csv')))
for r in rows[1:]:
    if '(anonymous namespace)::' in r[0] and ('k_fit' in r[0] or 'k_r' in r[0]):
        name = r[0].replace('(anonymous namespace)::', '').split('(')[0]
        print("%-28s calls %4s avg_ms %8.4f total_ms %9.3f" % (name, r[1], float(r[3]) / 1e6, float(r[2]) / 1e6))
PY
rm -f gpurun_out/wfks/wf_kernel_trace.csv
