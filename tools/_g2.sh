#!/bin/bash
rm -rf gpurun_out/r04_d
bash tools/profile_round.sh r04_d > /dev/null 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/profile_publish.py r04_d > /dev/null 2>&1
timeout 900 python bench.py > gpurun_out/r04_d/bench_default.json 2> gpurun_out/r04_d/bench_default.err; tail -c 200 gpurun_out/r04_d/bench_default.err; head -c 260 gpurun_out/r04_d/bench_default.json
rm -rf gpurun_out/ks
