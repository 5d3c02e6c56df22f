#!/bin/bash
# r4-26: small-batch kernel with the staging wave
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r426}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
PBL_BENCH_SHAPES=13824x5120:0.8,5120x13824:0.8 PBL_BENCH_MS=32 PBL_SB_WAVES=1024,1536,2048,3072 timeout 800 python tools/bench_small.py 2>&1 | tee $O/small.jsonl | cut -c1-1000
for w in 1536 2048 3072; do PBL_SB_WAVES_DEFAULT=$w timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 --small-batch-image 1 > $O/cfg4_$w.json 2> $O/cfg4_$w.err; echo w=$w $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*' $O/cfg4_$w.json | tr '\n' ' '); done
