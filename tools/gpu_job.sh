#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3s}; mkdir -p $O
python __graft_entry__.py > $O/build.txt 2>&1
for v in default v3; do
  if [ $v = default ]; then L=""; else L="build/libpbl_$v.so"; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
             "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_CACHE_MISS TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN" ; do
    i=$((i+1))
    PBL_LIB=$L PBL_BENCH_ONLY=fused PBL_BENCH_SHAPES=4096x4096:0.95 PBL_BENCH_PREHEAT_S=0.2 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/prof_$v/gemm_pmc$i -o pmc -- python tools/bench_gemm.py > $O/pmc_${v}_$i.log 2>&1
  done
  python tools/summarize_prof.py $O/prof_$v > $O/pmc_summary_$v.txt 2>&1
  rm -rf $O/prof_$v
done
cat $O/pmc_summary_*.txt | cut -c1-1200
