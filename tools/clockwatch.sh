#!/bin/bash
# Sample GPU clocks / power while bench.py runs a long sustained stream, for each ablation.
SRC="pb_llm_amd/csrc/pbl_kernels.hip pb_llm_amd/csrc/pbl_host.cpp"
cp pb_llm_amd/libpbl.so /tmp/libpbl_orig.so
for A in 0 1 2; do
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -shared --offload-arch=gfx950 -DPBL_ABLATE=$A $SRC -o pb_llm_amd/libpbl.so 2>/dev/null
  touch pb_llm_amd/libpbl.so
  echo "== PBL_ABLATE=$A"
  python bench.py --steps 30000 --warmup 10 --no-cpu-baseline > /tmp/b$A.log 2>&1 &
  BP=$!
  sleep 22
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ';'; echo; sleep 1
  done
  wait $BP
  tail -1 /tmp/b$A.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  layer-tok/s %.0f  GB/s %.0f  us/layer %.3f' % (j['value'], j['roofline']['achieved'], j['roofline']['us_per_layer']))"
done
cp /tmp/libpbl_orig.so pb_llm_amd/libpbl.so
