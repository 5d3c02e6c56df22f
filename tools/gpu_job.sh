#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3i}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -5 > $O/test_gemm.txt
timeout 600 python tools/bench_gemm.py > $O/bench_gemm.jsonl 2> $O/bench_gemm.err
MODES=prefill timeout 900 python tools/bench_llama7b.py > $O/llama7b_prefill.json 2> $O/llama7b_prefill.err
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/test_all.txt
cat $O/test_gemm.txt $O/bench_gemm.jsonl $O/llama7b_prefill.json $O/test_all.txt
