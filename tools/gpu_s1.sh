set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -x --deselect tests/test_gpu_configs.py::test_config5_ksplit_p2p_all_reduce_two_processes > gpurun_out/s1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
tail -40 gpurun_out/s1/pytest.log
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -k p2p > gpurun_out/s1/p2p.log 2>&1; echo "p2p rc=$?" >> gpurun_out/s1/p2p.log; tail -15 gpurun_out/s1/p2p.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench_driver.json 2> gpurun_out/s1/bench_driver.err; tail -c 1500 gpurun_out/s1/bench_driver.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --preheat-s 0 --no-cpu-baseline > gpurun_out/s1/bench_nopreheat.json 2>&1; tail -c 600 gpurun_out/s1/bench_nopreheat.json
PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9" timeout 300 python tools/bench_mfma.py > gpurun_out/s1/mfma.json 2>&1; cat gpurun_out/s1/mfma.json | tail -3
