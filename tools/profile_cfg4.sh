#!/bin/bash
# Profile `bench.py --workload cfg4` (BASELINE.json configs[3]: the small-batch kernel over the GEMM images): kernel trace + PMC passes,
# each in its own run.  Raw output -> gpurun_out/prof_<tag>/, summary -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-cfg4}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --workload cfg4 --steps 20 --warmup 5 $*"
rocprofv3 --kernel-trace --stats -d $OUT/sbimg_trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
  --kernel-trace -d $OUT/sbimg_pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  --kernel-trace -d $OUT/sbimg_pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/sbimg_fetch -o pmc -- $CMD > $OUT/fetch.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -size +8M -delete
cat $OUT/summary.txt
