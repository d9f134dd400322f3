export TMPDIR=/tmp
OUT=gpurun_out/r04_n; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_tables.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest.txt
Q="--steps 10 --warmup 2 --cpu-samples 0 --stage-inputs 0 --workflow-reps 0 --config1-steps 0 --fit-concordance 0 --verify-columns 4 --emit-mode tables-sm"
for v in "" "--counts-layout 1"; do
  timeout 300 python bench.py $Q $v > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print('[$v]', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()}, d['roofline']['kernel_ms_alone'], {k:v for k,v in d['verify'].items() if k!='what'})" || tail -5 $OUT/b.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python bench.py $Q --counts-layout 1 --verify-columns 0 --kernel-alone 0 > $OUT/ks.log 2>&1
python - <<PY
import csv,re
for r in csv.DictReader(open('$OUT/ks_kernel_stats.csv')):
    n=r['Name']
    if 'k_' in n and 'rocprim' not in n and int(r['Calls'])>2:
        print(re.search(r'(k_\w+)',n).group(1), r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
rm -f $OUT/*_kernel_trace.csv $OUT/*agent_info.csv $OUT/*domain_stats.csv
