#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Plumbing run of the multi-rank bench path on a one-GPU box: 2 ranks share cuda:0, control tensors over gloo, the data-path
# sums through the peer-to-peer all-reduce; per-layer (64 dependent reductions) and stacked.  Never a measurement.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3z}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
for tc in per-layer stacked; do
  PBL_BENCH_BACKEND=gloo timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 5 --warmup 2 --preheat-s 0.3 --collective p2p --tp-collectives $tc --no-cpu-baseline > $O/tp2_$tc.json 2> $O/tp2_$tc.err
  echo "rc=$?" >> $O/tp2_$tc.json
  tail -2 $O/tp2_$tc.json | cut -c1-900
done
tail -5 $O/tp2_per-layer.err | cut -c1-300
