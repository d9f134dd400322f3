set -x
mkdir -p gpurun_out/r6a
for d in 25 100 400 1600; do
  python bench.py --depth $d --steps 10 --warmup 2 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --regimes 0 --dropin 0 --config1-steps 0 --cpu-samples 0 --fit-concordance 0 --verify-columns 2 > gpurun_out/r6a/depth_$d.json 2> gpurun_out/r6a/depth_$d.err
done
for s in 64 256; do
  python bench.py --samples $s --steps 10 --warmup 2 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --regimes 0 --dropin 0 --config1-steps 0 --cpu-samples 0 --fit-concordance 0 --verify-columns 2 > gpurun_out/r6a/S_$s.json 2> gpurun_out/r6a/S_$s.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6a/*.json')):
    try:
        j=json.load(open(f))
        print(f, round(j['ms_per_step'],3), j['table_stats'], j['roofline']['kernel_ms'], j['roofline']['kernel_ms_alone'], j['stage_ms'], {k:j['verify'][k] for k in j['verify'] if k!='what'})
    except Exception as e:
        print(f,'ERR',e)
PY
