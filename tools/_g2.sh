export TMPDIR=/tmp
OUT=gpurun_out/r04_i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_tables.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
Q="--steps 6 --warmup 1 --cpu-samples 0 --stage-inputs 0 --workflow-reps 0 --config1-steps 0 --fit-concordance 0 --verify-columns 4 --emit-mode tables-sm"
timeout 300 python bench.py $Q > $OUT/b.json 2> $OUT/b.err
python -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()}, d['roofline']['kernel_ms_alone'], d['verify'])" || tail -5 $OUT/b.err
