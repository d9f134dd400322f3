#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_tables.py tests/test_shim.py -x -q 2>&1 | tail -4
timeout 900 python tools/fuzz_tables.py 420 31337 2>&1 | tail -2
