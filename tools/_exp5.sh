for rep in 1 2 3; do for v in "" head; do
python bench.py --steps 20 --warmup 5 --cpu-samples 0 ${v:+--lib-variant $v} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['ms_per_step'],3), round(d['roofline'].get('kernel_ms'),3))"
done; done
