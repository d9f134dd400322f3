mkdir -p gpurun_out/r04_a
python -m pytest tests/test_gpu_tables.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_a/pytest.txt
Q="--steps 10 --cpu-samples 0 --stage-inputs 0 --workflow-reps 0 --config1-steps 0 --fit-concordance 0 --verify-columns 4"
python bench.py $Q > gpurun_out/r04_a/strict.json 2> gpurun_out/r04_a/strict.err
for tw in 16 32 64; do ED_TAB_TW=$tw python bench.py $Q --emit-mode tables > gpurun_out/r04_a/tables_$tw.json 2> gpurun_out/r04_a/tables_$tw.err; done
cat gpurun_out/r04_a/pytest.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_a/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['stage_ms'], d['roofline']['kernel_ms_alone'], d['verify'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
