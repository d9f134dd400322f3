#!/bin/bash
# the round's closing sweeps on the final library
mkdir -p gpurun_out/r04_z
( timeout 400 python tools/fuzz_tables.py 200 777 2>&1 | tail -2
  timeout 400 python tools/fuzz_parity.py 150 4242 2>&1 | tail -2
  timeout 400 python tools/fuzz_cohort.py 150 99 2>&1 | tail -3
  timeout 400 python tools/fuzz_more.py 120 31 2>&1 | tail -3 ) > gpurun_out/r04_z/fuzz.txt 2>&1
cat gpurun_out/r04_z/fuzz.txt
