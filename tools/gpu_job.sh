#!/bin/bash
# What a gpurun call of this round typically ran (rewritten per call; this is the end-of-round validation, call r4-40):
# full GPU suite, smoke, the driver's bench command, the two side workloads.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-final}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'P' $O/bench_driver.json
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d["roofline"]; print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"],4), "frac", round(r["frac"],4), "sustained_frac", r.get("sustained_frac"), "cpu", d.get("cpu_baseline",{}).get("value"))
P
timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4.json 2> $O/cfg4.err; echo cfg4 $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*' $O/cfg4.json | tr '\n' ' ')
timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 3 > $O/cfg3.json 2> $O/cfg3.err; echo cfg3 $(grep -o '"us_per_step": [0-9.]*\|"frac": [0-9.]*' $O/cfg3.json | tr '\n' ' ')
