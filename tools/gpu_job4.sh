#!/bin/bash
# gpurun call 4 of round 5: the K-split tail of the image GEMM -- full GPU suite, then the shapes it is for (llama-13b at 2048 rows,
# short prompts), cfg3, the 7B-shaped forward.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r54}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt | cut -c1-700
PBL_BENCH_SHAPES=5120x5120:0.95,5120x13824:0.95,13824x5120:0.95,11008x4096:0.95 timeout 600 python tools/bench_gemm.py > $O/gemm_13b.jsonl 2> $O/gemm_13b.err; cut -c1-420 $O/gemm_13b.jsonl; tail -2 $O/gemm_13b.err | cut -c1-300
for m in 300 512 1024; do
  PBL_BENCH_M=$m PBL_BENCH_SHAPES=4096x4096:0.95,4096x11008:0.95,11008x4096:0.95 timeout 400 python tools/bench_gemm.py > $O/gemm_m$m.jsonl 2> $O/gemm_m$m.err; cut -c1-420 $O/gemm_m$m.jsonl
done
timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 3 > $O/cfg3.json 2> $O/cfg3.err; echo cfg3 $(grep -o '"us_per_step": [0-9.]*\|"frac": [0-9.]*' $O/cfg3.json | tr '\n' ' ')
MODES=prefill BF16=1 timeout 600 python tools/bench_llama7b.py > $O/llama7b.json 2> $O/llama7b.err; tail -1 $O/llama7b.json | cut -c1-1200
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; grep -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/bench_driver.json | head -3 | tr '\n' ' '
