#!/bin/bash
# Diagnostic PMC passes on the emission kernel: instruction cache, issue mix, LDS.   tools/pmc_diag.sh <tag>
set -u
TAG=${1:-pmcd}; OUT=gpurun_out/$TAG; export TMPDIR=/tmp; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --cpu-samples 0 --kernel-alone 0 --verify-columns 0 --fit-concordance 0 --config1-steps 0 --stage-inputs 0 --workflow-reps 0"
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o d$i -- $B > $OUT/d$i.log 2>&1
  python tools/pmc_summary.py $OUT/d${i}_counter_collection.csv | grep "kernel,\|k_emit_batch" > $OUT/diag$i.csv
  rm -f $OUT/d${i}_counter_collection.csv $OUT/d${i}_kernel_trace.csv $OUT/*agent_info.csv
  cat $OUT/diag$i.csv
done
