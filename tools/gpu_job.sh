#!/bin/bash
# One gpurun call (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Targeted re-check after the last edit of the unpack store loop: device unpack vs host unpack, GEMM regime, checkpoints.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r4d}; mkdir -p $O
timeout 60 python __graft_entry__.py > $O/build.txt 2>&1
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_pack.py -q -m gpu -k "unpack or gemm_regime or checkpoint_roundtrip or from_dense or config3" 2>&1 | tail -5 > $O/test_sel.txt
cat $O/test_sel.txt
