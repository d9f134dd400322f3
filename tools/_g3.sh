export TMPDIR=/tmp
OUT=gpurun_out/r04_l; mkdir -p $OUT
Q="--steps 10 --warmup 2 --cpu-samples 0 --stage-inputs 0 --workflow-reps 0 --config1-steps 0 --fit-concordance 0 --verify-columns 0 --emit-mode tables-sm"
for v in "" "--tables-early 1" "--tables-early 1 --split 0.5" "--tables-early 1 --split 0" "--batches-in-flight 3 --tables-early 1"; do
  timeout 300 python bench.py $Q $v > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print('[$v]', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()}, d['roofline']['kernel_ms_alone'])" || tail -5 $OUT/b.err
done
