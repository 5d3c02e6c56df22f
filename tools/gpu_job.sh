#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Small-batch matrix-core kernel with requests two slabs ahead: parity first, then timing against the previous build.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3y}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -q -m gpu -x -k "mfma or m32 or small_batch or random_layer or routing or shapes_and_batches" 2>&1 | tail -8 > $O/test_sel.txt
cat $O/test_sel.txt
for v in default oldm default oldm; do
  if [ $v = default ]; then L=""; else L="build/libpbl_$v.so"; fi
  echo -n "$v " >> $O/bench_mfma.txt
  PBL_LIB=$L PBL_BENCH_M=32,24 timeout 200 python tools/bench_mfma.py 2>/dev/null | tail -1 >> $O/bench_mfma.txt
done
cat $O/bench_mfma.txt
