#!/bin/bash
SRC="pb_llm_amd/csrc/pbl_kernels.hip pb_llm_amd/csrc/pbl_gemm.hip pb_llm_amd/csrc/pbl_qat.hip pb_llm_amd/csrc/pbl_prep.hip pb_llm_amd/csrc/pbl_host.cpp"
cp pb_llm_amd/libpbl.so /tmp/libpbl_orig.so
for W in 4 5 6 7 8; do
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -shared --offload-arch=gfx950 -DPBL_MIN_WAVES=$W $SRC -o pb_llm_amd/libpbl.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "ILi1ELi4E" | grep -E "VGPRs:|Scratch" | tr '\n' ' '
  touch pb_llm_amd/libpbl.so
  echo "== MIN_WAVES=$W"
  python bench.py --steps 3000 --warmup 500 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  layer-tok/s %.0f  GB/s %.0f  us/layer %.3f' % (j['value'], j['roofline']['achieved'], j['roofline']['us_per_layer']))"
done
cp /tmp/libpbl_orig.so pb_llm_amd/libpbl.so
