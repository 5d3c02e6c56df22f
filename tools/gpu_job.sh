#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Where do the ~20 us of the GEMM-regime kernel outside its loop go?  Kernel-trace durations of component builds.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r4a}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
for v in default e256 e512 loop0e loop0; do
  if [ $v = default ]; then L=""; else L="build/libpbl_$v.so"; fi
  PBL_LIB=$L PBL_BENCH_ONLY=fused PBL_BENCH_SHAPES=4096x4096:0.95 PBL_BENCH_PREHEAT_S=0.3 timeout 90 rocprofv3 --kernel-trace --stats -d $O/prof/${v}_trace -o trace -- python tools/bench_gemm.py > $O/$v.log 2>&1
done
timeout 60 python tools/summarize_prof.py $O/prof > $O/summary.txt 2>&1
rm -rf $O/prof
grep -A3 "_trace:" $O/summary.txt | cut -c1-260
