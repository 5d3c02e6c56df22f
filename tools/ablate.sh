#!/bin/bash
# Build ablated variants of libpbl.so (performance analysis only) and bench each.
# usage (on the GPU box): tools/ablate.sh [bench args]
set -u
SRC="pb_llm_amd/csrc/pbl_kernels.hip pb_llm_amd/csrc/pbl_gemm.hip pb_llm_amd/csrc/pbl_qat.hip pb_llm_amd/csrc/pbl_prep.hip pb_llm_amd/csrc/pbl_host.cpp"
cp pb_llm_amd/libpbl.so /tmp/libpbl_orig.so
for A in 0 1 2; do
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -shared --offload-arch=gfx950 -DPBL_ABLATE=$A $SRC -o pb_llm_amd/libpbl.so 2>/dev/null
  touch pb_llm_amd/libpbl.so
  echo "== PBL_ABLATE=$A (0 full, 1 memory only, 2 compute only)"
  python bench.py --steps 3000 --warmup 500 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  layer-tok/s %.0f  GB/s %.0f  us/layer %.3f' % (j['value'], j['roofline']['achieved'], j['roofline']['us_per_layer']))"
done
cp /tmp/libpbl_orig.so pb_llm_amd/libpbl.so
