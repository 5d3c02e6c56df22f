#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3l}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -5 > $O/test_gemv.txt
for v in default ring2 ring3w7 ring3w8 default ring2; do
  echo "== $v" >> $O/bench_gemv.txt
  if [ $v = default ]; then L=""; else L="build/libpbl_$v.so"; fi
  PBL_LIB=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['us_per_launch'],1))" >> $O/bench_gemv.txt
done
timeout 300 python tools/bench_p2p.py > $O/bench_p2p.json 2> $O/bench_p2p.err
cat $O/test_gemv.txt $O/bench_gemv.txt $O/bench_p2p.json
