export TMPDIR=/tmp
OUT=gpurun_out/r04_c; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_tables.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
Q="--steps 6 --warmup 1 --cpu-samples 0 --stage-inputs 0 --workflow-reps 0 --config1-steps 0 --fit-concordance 0 --verify-columns 0 --emit-mode tables"
for tw in 4 8 16; do
  ED_TAB_TW=$tw timeout 200 python bench.py $Q > $OUT/b_$tw.json 2> $OUT/b_$tw.err
  python -c "
import json;d=json.loads(open('$OUT/b_$tw.json').read().strip().splitlines()[-1]);print('TW$tw', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()}, d['roofline']['kernel_ms_alone'])"
done
for tw in 8 16; do
  ED_TAB_TW=$tw timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks$tw -- python bench.py $Q --kernel-alone 0 --fit 0 --pipeline 0 > $OUT/ks$tw.log 2>&1
  echo "== TW$tw kernel stats (fit 0, pipeline 0)"; grep "k_" $OUT/ks${tw}_kernel_stats.csv | cut -c1-160 | head -14
  ED_TAB_TW=$tw timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT -o p$tw -- python bench.py $Q --kernel-alone 0 --fit 0 --pipeline 0 --steps 2 > $OUT/p$tw.log 2>&1
  python tools/pmc_summary.py $OUT/p${tw}_counter_collection.csv | grep "k_emit_tab"
  rm -f $OUT/*_counter_collection.csv $OUT/*_kernel_trace.csv $OUT/*agent_info.csv $OUT/*domain_stats.csv
done
