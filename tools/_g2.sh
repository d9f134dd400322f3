#!/bin/bash
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
mkdir -p gpurun_out/r04_b; timeout 900 python bench.py > gpurun_out/r04_b/bench_default.json 2> gpurun_out/r04_b/bench_default.err; tail -c 600 gpurun_out/r04_b/bench_default.err; head -c 400 gpurun_out/r04_b/bench_default.json
