set -u
export TMPDIR=/tmp
O=gpurun_out/s29; mkdir -p $O
export PBL_BENCH_CACHE=/tmp/c6.pt PBL_BENCH_SHAPES="4096x4096:0.9,13824x5120:0.8,11008x4096:0.95,4096x11008:0.9" PBL_BENCH_M=8,16,32
for v in base pf2; do
  if [ $v = base ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
  echo -n "$v "; python tools/bench_mfma.py 2>&1 | tail -1
done | tee $O/pf2.txt
PBL_LIB=build/libpbl_pf2.so timeout 600 python -m pytest -m gpu -q tests/test_gpu_fuzz.py tests/test_gpu_groups.py 2>&1 | tail -2
unset PBL_LIB
timeout 600 python -m pytest -m gpu -q tests/test_gpu_fuzz.py tests/test_gpu_parity.py -k "mfma or fuzz or random" 2>&1 | tail -2
