export TMPDIR=/tmp
for d in 50 100; do
  rm -rf gpurun_out/pm$d; mkdir -p gpurun_out/pm$d
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pm$d -o p -- python bench.py --steps 2 --warmup 1 --cpu-samples 0 --depth $d > gpurun_out/pm$d/log 2>&1
  python tools/pmc_summary.py gpurun_out/pm$d/p_counter_collection.csv | grep -E "k_fit_hnewton|k_fit_hist"
  rm -f gpurun_out/pm$d/p_counter_collection.csv gpurun_out/pm$d/p_kernel_trace.csv
done
