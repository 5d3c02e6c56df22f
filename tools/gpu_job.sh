#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3c}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -8 > $O/test_gemm.txt
timeout 600 python tools/bench_gemm.py > $O/bench_gemm.jsonl 2> $O/bench_gemm.err
PBL_BENCH_ONLY=fused PBL_BENCH_SHAPES=4096x4096:0.9 timeout 200 python tools/bench_gemm.py >> $O/bench_gemm.jsonl 2>> $O/bench_gemm.err
export PBL_BENCH_ONLY=fused PBL_BENCH_SHAPES=4096x4096:0.95 PBL_BENCH_PREHEAT_S=0.5
for v in abl1 abl2 abl35 abl39 abl4; do
  echo "== $v" >> $O/bench_variants.txt
  PBL_LIB=build/libpbl_$v.so timeout 200 python tools/bench_gemm.py >> $O/bench_variants.txt 2>> $O/bench_variants.err
done
# exact kernel durations (the event timing of a 20 us kernel is host bound) and the PMC picture of the consumer loop alone
for v in abl35 abl39; do
  PBL_LIB=build/libpbl_$v.so rocprofv3 --kernel-trace --stats -d $O/gemm_${v}_trace -o trace -- python tools/bench_gemm.py > $O/gemm_${v}_trace.log 2>&1
done
PBL_LIB=build/libpbl_abl35.so rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace -d $O/gemm_abl35_pmc2 -o pmc -- python tools/bench_gemm.py > $O/gemm_abl35_pmc2.log 2>&1
PBL_LIB=build/libpbl_abl35.so rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --kernel-trace -d $O/gemm_abl35_pmc1 -o pmc -- python tools/bench_gemm.py > $O/gemm_abl35_pmc1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/gemm_full_trace -o trace -- python tools/bench_gemm.py > $O/gemm_full_trace.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace -d $O/gemm_full_pmc2 -o pmc -- python tools/bench_gemm.py > $O/gemm_full_pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --kernel-trace -d $O/gemm_full_pmc1 -o pmc -- python tools/bench_gemm.py > $O/gemm_full_pmc1.log 2>&1
python tools/summarize_prof.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
cat $O/test_gemm.txt $O/bench_gemm.jsonl $O/bench_variants.txt; cut -c1-700 $O/summary.txt
