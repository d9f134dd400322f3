#!/bin/bash
timeout 300 python tools/_g3.py 2>&1 | grep "7 max\|oracle" | head -6
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fit_concordance.py tests/test_gpu_refcohort.py tests/test_gpu_refset.py tests/test_gpu_config1.py -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_fit_sm.py 150 2>&1 | tail -4
