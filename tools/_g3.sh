export TMPDIR=/tmp
OUT=gpurun_out/r04_q; mkdir -p $OUT
timeout 900 python bench.py > $OUT/default.json 2> $OUT/default.err
tail -3 $OUT/default.err
python - <<PY
import json
d=json.loads(open('$OUT/default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config'])
print(d['roofline'])
print(d['stage_ms'])
print(d['verify'])
print(d['fit_concordance'])
print(d['extra'])
print(d.get('value_with_h2d'), d.get('h2d'))
print({k:v for k,v in d['cpu_baseline'].items() if k!='all_cores'})
PY
