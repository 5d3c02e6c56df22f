#!/bin/bash
# r4-33: final small-batch kernel; from how many rows it beats the library's own routing (GEMV passes / records kernel)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r433}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
PBL_BENCH_SHAPES=13824x5120:0.8,11008x4096:0.9,4096x4096:0.9 PBL_BENCH_MS=2,3,4,5,6,8 PBL_SB_WAVES=0 timeout 800 python tools/bench_small.py 2>&1 | tee $O/small.jsonl | grep -v amdgpu.ids | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 --small-batch-image 1 > $O/cfg4.json 2> $O/cfg4.err; echo $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*\|"image_bytes": [0-9]*' $O/cfg4.json | tr '\n' ' ')
