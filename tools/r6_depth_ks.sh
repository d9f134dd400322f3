# kernel stats of the default line at depths 100 / 400 / 1600 (round 6: where does the depth cliff come from?)
for d in 100 400 1600; do
  echo "== depth $d"; bash tools/kernel_stats.sh --depth $d; cp gpurun_out/ks/ks_kernel_stats.csv gpurun_out/r6a/ks_depth_$d.csv; grep -h '^{' gpurun_out/ks/log > gpurun_out/r6a/ks_line_depth_$d.json
done
