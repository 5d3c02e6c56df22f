#!/bin/bash
# r4-36: experiment: in-kernel reduce with 16-byte system-scope stores / loads
set -u
export TMPDIR=/tmp
PBL_BENCH_SHAPES=13824x5120:0.8,5120x13824:0.8,11008x4096:0.9 PBL_BENCH_MS=32 PBL_SB_WAVES=0 PBL_SB_FLAGS=4 timeout 800 python tools/bench_small.py 2>&1 | grep -v amdgpu.ids | cut -c1-500
