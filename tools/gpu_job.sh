#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Timing-only probes of the headline GEMV (PBL_PROBE builds: results wrong by construction, only the launch time is read).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3w}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
for rep in 1 2; do
for v in default p1 p2 p4 p8 p16 p7; do
  if [ $v = default ]; then L=""; else L="build/libpbl_$v.so"; fi
  echo -n "$v " >> $O/bench_gemv.txt
  PBL_LIB=$L timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['us_per_launch'],1))" >> $O/bench_gemv.txt
done; done
cat $O/bench_gemv.txt
