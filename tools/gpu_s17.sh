set -u
export TMPDIR=/tmp
O=gpurun_out/s17; mkdir -p $O
export PBL_BENCH_SHAPES="13824x5120:0.8" PBL_BENCH_M=32
for v in a32 a64 a96; do
  export PBL_LIB=build/libpbl_$v.so
  echo -n "$v " ; timeout 600 python tools/bench_mfma.py 2>&1 | tail -1
done | tee $O/ablate.txt
