set -u
export TMPDIR=/tmp
O=gpurun_out/s23; mkdir -p $O
export PBL_BENCH_CACHE=/tmp/c6.pt PBL_BENCH_SHAPES="4096x4096:0.9,13824x5120:0.8,11008x4096:0.95,4096x11008:0.9" PBL_BENCH_M=8,32
for v in base s256 s768 s1024 m4 m1; do
  if [ $v = base ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
  echo -n "$v "; python tools/bench_mfma.py 2>&1 | tail -1
done | tee $O/split.txt
