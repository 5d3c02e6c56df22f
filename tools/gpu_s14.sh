set -u
export TMPDIR=/tmp
O=gpurun_out/s14; mkdir -p $O
export PBL_BENCH_SHAPES="13824x5120:0.8" PBL_BENCH_M=32,16
for v in base a2 a4 a6 a14 a30; do
  if [ $v = base ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
  echo -n "$v " ; timeout 600 python tools/bench_mfma.py 2>&1 | tail -1
done | tee $O/ablate.txt
