#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -3
