set -u
export TMPDIR=/tmp
O=gpurun_out/s30; mkdir -p $O
PBL_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --collective p2p --no-cpu-baseline > $O/tp2_gloo_p2p.json 2>$O/tp2.err; tail -1 $O/tp2_gloo_p2p.json | cut -c1-700; tail -3 $O/tp2.err
timeout 600 python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-600 | tee $O/cfg4.json
timeout 900 python bench.py --workload cfg3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-600 | tee $O/cfg3.json
