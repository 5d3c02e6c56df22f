#!/bin/bash
# One gpurun call of round 3 (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# Round-end validation of the committed state: full -m gpu suite, the driver's bench command, smoke, the side workloads.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r3x}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/test_all.txt
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
for b in library fused; do timeout 100 python bench.py --workload cfg3 --gemm-backend $b --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_side.jsonl; done
timeout 100 python bench.py --workload cfg4 --steps 50 --warmup 10 2>/dev/null | tail -1 >> $O/bench_side.jsonl
MODES=prefill timeout 300 python tools/bench_llama7b.py > $O/llama7b_prefill.json 2> $O/llama7b.err
cat $O/test_all.txt; tail -1 $O/bench_driver.json | cut -c1-330; tail -2 $O/smoke.txt; cut -c1-420 $O/bench_side.jsonl; tail -1 $O/llama7b_prefill.json | cut -c1-900
