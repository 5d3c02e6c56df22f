#!/bin/bash
# gpurun call 3 of round 5: diagnostics of the next-layer prefetch (negative so far), the 7B-shaped forward (fp16 + bf16), and the
# rocprofv3 profiles of the driver's bench command, of cfg4 and of cfg3.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r53}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
# (a) upper bound of what warm caches can give the sequential-decode pattern: 8 layers (50 MB: inside the Infinity Cache) vs 224
timeout 200 python bench.py --mode graph --prefetch-next 0 --layers 8 --steps 200 --warmup 20 --no-cpu-baseline > $O/graph_l8.json 2> $O/graph_l8.err
echo "graph 8 layers (cache resident)" $(grep -o '"us_per_layer": [0-9.]*' $O/graph_l8.json | head -1)
timeout 200 python bench.py --mode graph --prefetch-next 0 --steps 50 --warmup 10 --no-cpu-baseline > $O/graph_pf0.json 2> $O/graph_pf0.err
echo "graph 224 layers pf=0" $(grep -o '"us_per_layer": [0-9.]*' $O/graph_pf0.json | head -1)
PBL_LIB=build/libpbl_pf8.so timeout 200 python bench.py --mode graph --prefetch-next 1 --steps 50 --warmup 10 --no-cpu-baseline > $O/graph_pf8.json 2> $O/graph_pf8.err
echo "graph 224 layers pf=1, 8 prefetch workgroups" $(grep -o '"us_per_layer": [0-9.]*' $O/graph_pf8.json | head -1); tail -1 $O/graph_pf8.err | cut -c1-200
# (b) the 7B-shaped forward
timeout 900 python tools/bench_llama7b.py > $O/llama7b.json 2> $O/llama7b.err; tail -1 $O/llama7b.json | cut -c1-1500; tail -2 $O/llama7b.err | cut -c1-300
# (c) profiles
PROF_STEPS=20 PROF_WARMUP=5 timeout 900 bash tools/profile.sh r05_final > $O/profile_final.txt 2>&1; tail -30 $O/profile_final.txt | cut -c1-400
timeout 600 bash tools/profile_cfg4.sh r05_cfg4 > $O/profile_cfg4.txt 2>&1; tail -12 $O/profile_cfg4.txt | cut -c1-400
mkdir -p gpurun_out/prof_r05_cfg3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05_cfg3/gemmimg_trace -o trace -- python bench.py --workload cfg3 --steps 10 --warmup 3 > gpurun_out/prof_r05_cfg3/trace.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r05_cfg3 > gpurun_out/prof_r05_cfg3/summary.txt 2>&1; cut -c1-300 gpurun_out/prof_r05_cfg3/summary.txt | head -20
find gpurun_out/prof_r05_final gpurun_out/prof_r05_cfg4 gpurun_out/prof_r05_cfg3 -name "*.db" -size +6M -delete
