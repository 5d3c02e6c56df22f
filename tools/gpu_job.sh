#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3k}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pack.py tests/test_gpu_gemm.py tests/test_gpu_parity.py -q -x 2>&1 | tail -12 > $O/test_a.txt
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -x -k "g10 or g9 or fused_members or fused_decode or p2p" 2>&1 | tail -12 > $O/test_b.txt
timeout 600 python tools/bench_host.py > $O/bench_host.json 2> $O/bench_host.err
MODES=decode timeout 900 python tools/bench_llama7b.py > $O/llama7b_decode.json 2> $O/llama7b_decode.err
PBL_NATIVE=0 MODES=decode timeout 900 python tools/bench_llama7b.py > $O/llama7b_decode_ctypes.json 2> $O/llama7b_decode_ctypes.err
cat $O/test_a.txt $O/test_b.txt $O/bench_host.json $O/llama7b_decode.json $O/llama7b_decode_ctypes.json
