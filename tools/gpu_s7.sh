set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s7
export PBL_BENCH_ONLY=fused PBL_BENCH_SHAPES="4096x4096:0.95"
for v in "$@"; do
  echo "== $v"; PBL_LIB=$PWD/build/libpbl_$v.so timeout 300 python tools/bench_gemm.py 2>&1 | grep shape | tee -a gpurun_out/s7/gemm_$v.json
done
