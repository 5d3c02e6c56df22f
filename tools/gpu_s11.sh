set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s11
timeout 900 python -m pytest tests/test_gpu_groups.py -m gpu -q > gpurun_out/s11/groups.log 2>&1; echo "rc=$?" >> gpurun_out/s11/groups.log; tail -40 gpurun_out/s11/groups.log
PBL_BENCH_CACHE=/tmp/c2.pt PBL_BENCH_SHAPES="4096x4096:0.9,4096x4096:0.9:128,4096x4096:0.9:1024" timeout 300 python tools/bench_mfma.py > gpurun_out/s11/mfma_groups.json 2>&1; tail -1 gpurun_out/s11/mfma_groups.json
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_groups.py > gpurun_out/s11/full.log 2>&1; echo "rc=$?" >> gpurun_out/s11/full.log; tail -15 gpurun_out/s11/full.log
