set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s12
timeout 900 python -m pytest tests/test_gpu_groups.py -m gpu -q > gpurun_out/s12/groups.log 2>&1; echo "rc=$?" >> gpurun_out/s12/groups.log; grep -n "^E  \|passed\|failed\|^FAILED" gpurun_out/s12/groups.log | head -40
