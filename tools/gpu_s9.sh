set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s9
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "fused_gemm or mfma or fuzz or small_batch or k_split or config3" > gpurun_out/s9/tests.log 2>&1; echo "rc=$?" >> gpurun_out/s9/tests.log; tail -6 gpurun_out/s9/tests.log
PBL_BENCH_ONLY=fused timeout 300 python tools/bench_gemm.py > gpurun_out/s9/gemm.json 2>&1; tail -3 gpurun_out/s9/gemm.json
PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9,4096x4096:0.95" timeout 300 python tools/bench_mfma.py > gpurun_out/s9/mfma.json 2>&1; tail -1 gpurun_out/s9/mfma.json
