set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s10
timeout 600 python -m pytest tests/test_gpu_pack.py -m gpu -q -x > gpurun_out/s10/pack_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s10/pack_tests.log; tail -30 gpurun_out/s10/pack_tests.log
timeout 300 python tools/bench_pack.py > gpurun_out/s10/pack_bench.json 2>&1; tail -3 gpurun_out/s10/pack_bench.json
PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9" timeout 300 python tools/bench_mfma.py > gpurun_out/s10/mfma.json 2>&1; tail -1 gpurun_out/s10/mfma.json
