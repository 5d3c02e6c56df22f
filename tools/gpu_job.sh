#!/bin/bash
# r4-28: the split rule (two workgroups per CU); cfg4 bench; llama shapes at 8 - 32 rows
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r428}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
PBL_BENCH_SHAPES=13824x5120:0.8,5120x13824:0.8,11008x4096:0.9,4096x11008:0.9,4096x4096:0.9 PBL_BENCH_MS=32,16,8 PBL_SB_WAVES=0 timeout 800 python tools/bench_small.py 2>&1 | tee $O/small.jsonl | cut -c1-400
timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 --small-batch-image 1 > $O/cfg4.json 2> $O/cfg4.err; echo $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*\|"image_bytes": [0-9]*' $O/cfg4.json | tr '\n' ' ')
