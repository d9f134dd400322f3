#!/bin/bash
bash tools/ab.sh tabper8 3 --strict-steps 0 2>&1 | tail -6
bash tools/kernel_stats.sh --lib-variant tabper8 2>&1 | grep "k_tab_build"
bash tools/kernel_stats.sh 2>&1 | grep "k_tab_build"
rm -rf gpurun_out/ks
