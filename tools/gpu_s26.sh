set -u
export TMPDIR=/tmp
timeout 600 python -m pytest -m gpu -q tests/test_gpu_fuzz.py tests/test_gpu_groups.py 2>&1 | grep -E "^E  |passed|failed" | head -10
PBL_BENCH_CACHE=/tmp/c2.pt PBL_BENCH_SHAPES="4096x4096:0.9:128,11008x4096:0.95:128" PBL_BENCH_M=8,16,32 timeout 300 python tools/bench_mfma.py 2>&1 | tail -1
