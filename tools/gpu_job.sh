#!/bin/bash
# r4-39: 3 working waves + stager per workgroup (4-wave workgroups place cleanly at any occupancy): ring 4 at 3 waves / SIMD, ring 2 at 4
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r439}; mkdir -p $O
for lib in pb_llm_amd/libpbl.so build/libpbl_w3a.so build/libpbl_w3b.so build/libpbl_w3c.so build/libpbl_w4c.so; do
  n=$(basename $lib .so)
  echo == $n $(PBL_LIB=$lib PBL_BENCH_SHAPES=13824x5120:0.8,11008x4096:0.9 PBL_BENCH_MS=32 PBL_SB_WAVES=0 timeout 400 python tools/bench_small.py 2>&1 | grep -o '"shape": "[0-9x]*"\|"image_us_w0": [0-9.]*\|"rel_err_w0": [0-9.e-]*' | tr '\n' ' ')
  PBL_LIB=$lib PBL_NATIVE=0 timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4_$n.json 2> $O/cfg4_$n.err; echo "   cfg4 (ctypes route)" $(grep -o '"us_per_layer": [0-9.]*' $O/cfg4_$n.json)
done
