#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Round-end evidence for the headline: rocprofv3 kernel trace of the driver's command + FETCH_SIZE / WRITE_SIZE passes.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3v}; P=$O/prof; mkdir -p $P
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
timeout 150 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- python bench.py $ARGS > $O/trace.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc3 -o pmc -- python bench.py $ARGS > $O/pmc3.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc4 -o pmc -- python bench.py $ARGS > $O/pmc4.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $P/pmc1 -o pmc -- python bench.py $ARGS > $O/pmc1.log 2>&1
timeout 60 python tools/summarize_prof.py $P > $O/summary.txt 2>&1
grep -h '^{"metric"' $O/trace.log $O/pmc3.log | cut -c1-200 >> $O/summary.txt
rm -rf $P
cat $O/summary.txt | cut -c1-600
