export TMPDIR=/tmp
OUT=gpurun_out/r04_u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tables.py tests/test_gpu_fit.py tests/test_gpu_fit_concordance.py tests/test_gpu_cohort.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -5
Q="--steps 10 --warmup 3 --cpu-samples 0 --stage-inputs 0 --workflow-reps 0 --config1-steps 0 --fit-concordance 0 --verify-columns 0 --strict-steps 0 --kernel-alone 0"
for v in "" "--tables-early 1" "--batches-in-flight 3" "--tables-early 1 --batches-in-flight 3"; do
  timeout 300 python bench.py $Q $v > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print('[$v]', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})" || tail -5 $OUT/b.err
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py $Q > $OUT/t.log 2>&1
python tools/trace_steps.py $OUT/t_kernel_trace.csv | head -30 | tee $OUT/timeline.txt
rm -f $OUT/t_kernel_trace.csv $OUT/*agent_info.csv
