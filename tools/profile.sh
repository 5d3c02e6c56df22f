#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + PMC passes (each in its own run, as the
# MI355X guide prescribes).  Raw output -> gpurun_out/prof_<tag>/, summary -> gpurun_out/prof_<tag>/summary.txt
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# PROF_STEPS / PROF_WARMUP: the command being profiled (the driver's is --steps 20 --warmup 5); the summary splits the kernel trace
# into the pre-heat loop and the last W + K dispatches (the timed region) with them
STEPS=${PROF_STEPS:-300}; WARM=${PROF_WARMUP:-100}
export PBL_PROF_WINDOW=$((STEPS + WARM))
ARGS="--steps $STEPS --warmup $WARM --no-cpu-baseline --no-side $*"      # (the side entries are profiled on their own: gpu_job6.sh profiles)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
  --kernel-trace -d $OUT/pmc1 -o pmc -- python bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  --kernel-trace -d $OUT/pmc2 -o pmc -- python bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc -- python bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4 -o pmc -- python bench.py $ARGS > $OUT/pmc4.log 2>&1
if [ -x build/calib_fetch ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/calib -o pmc -- ./build/calib_fetch > $OUT/calib.log 2>&1
fi
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
