set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s8
# multi-rank plumbing of bench.py on one device (gloo control plane, p2p data plane)
PBL_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --preheat-s 0.5 --collective p2p --layers 56 > gpurun_out/s8/bench_tp2_plumbing.json 2> gpurun_out/s8/bench_tp2.err; echo "rc=$?"; tail -c 900 gpurun_out/s8/bench_tp2_plumbing.json; tail -5 gpurun_out/s8/bench_tp2.err
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --durations=25 > gpurun_out/s8/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s8/pytest.log; tail -45 gpurun_out/s8/pytest.log
