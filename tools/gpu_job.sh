#!/bin/bash
# One gpurun call (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# pbl_unpack_dev with and without non-temporal stores, alone and followed by the library GEMM that reads its output.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r4c}; mkdir -p $O
timeout 60 python __graft_entry__.py > $O/build.txt 2>&1
PBL_BENCH_SHAPES=13824x5120:0.8,4096x4096:0.9 PBL_BENCH_M=32 timeout 120 python tools/bench_mfma.py > $O/mfma_cache.txt 2>&1
for v in default unt default unt; do
  if [ $v = default ]; then L=""; else L="build/libpbl_$v.so"; fi
  echo -n "$v " >> $O/unpack.txt
  PBL_LIB=$L timeout 60 python tools/bench_unpack.py 2>/dev/null | tail -1 >> $O/unpack.txt
  echo -n "$v " >> $O/gemm_lib.txt
  PBL_LIB=$L PBL_BENCH_ONLY=library PBL_BENCH_SHAPES=4096x4096:0.95 PBL_BENCH_PREHEAT_S=0.5 timeout 60 python tools/bench_gemm.py 2>/dev/null | tail -1 >> $O/gemm_lib.txt
done
cat $O/unpack.txt $O/gemm_lib.txt | cut -c1-400
