#!/bin/bash
# gpurun call 2 of round 5: full GPU suite (in-kernel bf16 GEMV, fused bf16 decode), the driver's bench command, the next-layer
# prefetch experiment (sequential-decode pattern: one launch per layer in a hipGraph, with / without pbl_linear_f16_pf), host cost.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r52}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt | cut -c1-600
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'P' $O/bench_driver.json
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d["roofline"]; print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"],4), "frac", round(r["frac"],4), "sustained_frac", r.get("sustained_frac"))
P
for pf in 0 1 0 1; do
  timeout 300 python bench.py --mode graph --prefetch-next $pf --steps 50 --warmup 10 --no-cpu-baseline > $O/graph_pf$pf.json 2> $O/graph_pf$pf.err
  echo graph prefetch=$pf $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*' $O/graph_pf$pf.json | head -2 | tr '\n' ' '); tail -2 $O/graph_pf$pf.err | cut -c1-200
done
timeout 300 python bench.py --mode eager --prefetch-next 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/eager_pf1.json 2> $O/eager_pf1.err; echo eager pf=1 $(grep -o '"us_per_layer": [0-9.]*' $O/eager_pf1.json | head -1)
timeout 300 python tools/bench_host.py > $O/host.json 2> $O/host.err; cut -c1-900 $O/host.json
