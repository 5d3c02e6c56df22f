set -u
export TMPDIR=/tmp
O=gpurun_out/s27; mkdir -p $O
timeout 600 python -m pytest -m gpu -q tests/test_gpu_groups.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -k "group or fuzz or random or zero or bf16 or routing" 2>&1 | grep -E "^E  |passed|failed" | head -10
export PBL_BENCH_CACHE=/tmp/c6.pt PBL_BENCH_SHAPES="4096x4096:0.9,13824x5120:0.8,11008x4096:0.95,4096x11008:0.9,1024x4096:0.9" PBL_BENCH_M=8,16
for v in base m3 m4 s640; do
  if [ $v = base ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
  echo -n "$v "; python tools/bench_mfma.py 2>&1 | tail -1
done | tee $O/split_x1.txt
