set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s2
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/s2/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s2/pytest.log
tail -60 gpurun_out/s2/pytest.log
PBL_BENCH_SHAPES="13824x5120:0.8,5120x13824:0.8,4096x4096:0.9" timeout 300 python tools/bench_mfma.py > gpurun_out/s2/mfma.json 2>&1; tail -3 gpurun_out/s2/mfma.json
