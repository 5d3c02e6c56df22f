#!/bin/bash
# r4-13: full GPU suite after the wait fix; GEMM timing again (the r49/r410 numbers were taken with the lax wait)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r413}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/test_all.txt
cat $O/test_all.txt
timeout 300 python tools/bench_gemm.py > $O/bench_gemm.jsonl 2> $O/bench_gemm.err; cat $O/bench_gemm.jsonl
PBL_BENCH_METRIC=hessian timeout 300 python tools/bench_gemm.py > $O/bench_gemm_h.jsonl 2> $O/bench_gemm_h.err; cat $O/bench_gemm_h.jsonl
