#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu --durations=5 2>&1 | tail -14
