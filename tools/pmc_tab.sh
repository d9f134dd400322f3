#!/bin/bash
# Diagnostic PMC passes on the table-driven emission kernel (cache behaviour of the gathers).   tools/pmc_tab.sh <tag> [extra bench flags]
set -u
TAG=${1:-pmct}; shift; OUT=gpurun_out/$TAG; export TMPDIR=/tmp; mkdir -p $OUT
B="python bench.py --emit-mode tables --steps 2 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0 $*"
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TCP\|TCC\|TA\|TD\)_[A-Z0-9_]*" | sort -u > $OUT/avail.txt
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o d$i -- $B > $OUT/d$i.log 2>&1
  python tools/pmc_summary.py $OUT/d${i}_counter_collection.csv 2>> $OUT/d$i.log | grep "kernel,\|k_emit_tab\|k_tab_\|k_viterbi" > $OUT/diag$i.csv
  rm -f $OUT/d${i}_counter_collection.csv $OUT/d${i}_kernel_trace.csv $OUT/*agent_info.csv
  cat $OUT/diag$i.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- $B > $OUT/ks.log 2>&1
grep "k_" $OUT/ks_kernel_stats.csv | head -30
rm -f $OUT/ks_kernel_trace.csv $OUT/ks_domain_stats.csv $OUT/*agent_info.csv
