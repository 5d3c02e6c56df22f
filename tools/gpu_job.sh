#!/bin/bash
# One gpurun call (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Round-end validation of the committed state: full -m gpu suite, the driver's bench command, smoke.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-final}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/test_all.txt
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
cat $O/test_all.txt; tail -1 $O/bench_driver.json | cut -c1-330; tail -2 $O/smoke.txt
