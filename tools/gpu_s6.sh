set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s6
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "fused_gemm" > gpurun_out/s6/gemm_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s6/gemm_tests.log; tail -5 gpurun_out/s6/gemm_tests.log
PBL_BENCH_ONLY=fused timeout 600 python tools/bench_gemm.py > gpurun_out/s6/gemm.json 2>&1; tail -4 gpurun_out/s6/gemm.json
