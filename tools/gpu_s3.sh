set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mfma" > gpurun_out/s3/mfma_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s3/mfma_tests.log; tail -15 gpurun_out/s3/mfma_tests.log
if grep -q "rc=0" gpurun_out/s3/mfma_tests.log; then
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/s3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s3/pytest.log; tail -30 gpurun_out/s3/pytest.log
  PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9" timeout 300 python tools/bench_mfma.py > gpurun_out/s3/mfma.json 2>&1; tail -2 gpurun_out/s3/mfma.json
fi
