set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s5
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "fused_gemm" > gpurun_out/s5/gemm_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s5/gemm_tests.log; tail -25 gpurun_out/s5/gemm_tests.log
timeout 600 python tools/bench_gemm.py > gpurun_out/s5/gemm.json 2>&1; tail -4 gpurun_out/s5/gemm.json
