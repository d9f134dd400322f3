#!/bin/bash
# One process against two processes sharing the GPU (two independent cohort pipelines), same box: ms per 200 000 x 1024 slab.   tools/ab_two_procs.sh [rounds]
N=${1:-2}
F="--cpu-samples 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 --strict-steps 0 --regimes 0 --dropin 0 --kernel-alone 0 --steps 20 --warmup 3"
for i in $(seq 1 $N); do
  python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one process ', round(d['ms_per_step'],3), 'ms per slab')"
  ED_BENCH_SHARE_GPU=1 ED_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2967$i bench.py --gpus 2 $F 2>/dev/null | \
    python -c "import json,sys; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]; print('two processes', round(d['ms_per_step']/2,3), 'ms per slab (', round(d['ms_per_step'],3), 'per step of two slabs )')"
done
