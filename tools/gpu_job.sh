#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "random_layers" 2>&1 | tail -12
