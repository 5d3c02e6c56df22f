#!/bin/bash
# r4-37: capture test of the small-batch image path, bf16 tests, GEMM suite after the cleanup
set -u
export TMPDIR=/tmp
timeout 120 python __graft_entry__.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "capture or bf16 or native or llama13b" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python tools/bench_host.py 2>&1 | tail -6 | cut -c1-300
