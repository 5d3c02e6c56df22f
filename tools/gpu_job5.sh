#!/bin/bash
# gpurun call 5 of round 5: the small-batch kernel with more working waves per workgroup (SB_WAVES = 5, 6, 7 instead of 4: more slot
# bytes in flight per CU at the same two workgroups per CU) on BASELINE configs[3] and per shape.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r55}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
for v in default sbw5 sbw6 sbw7; do
  if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
  timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4_$v.json 2> $O/cfg4_$v.err
  echo cfg4 $v $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*' $O/cfg4_$v.json | tr '\n' ' '); tail -1 $O/cfg4_$v.err | cut -c1-200
  PBL_BENCH_MS=32,8 timeout 400 python tools/bench_small.py > $O/small_$v.jsonl 2> $O/small_$v.err
  python - <<P $O/small_$v.jsonl
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("  ", d["shape"], "M", d["M"], "image_us", d.get("image_us_w2048"), "records_us", d.get("records_us"), "dense_us", d.get("dense_us"), "err", d.get("rel_err_w2048"), "repeat", d.get("repeat_w2048"))
P
done
