#!/bin/bash
# end-of-round validation: full GPU suite (twice: flakiness), smoke, the driver's bench command, cfg3 / cfg4 with library defaults.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r56}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
for i in 1 2; do
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest$i.txt 2>&1; tail -25 $O/pytest$i.txt | grep -v "^$" | cut -c1-600 | tail -14
done
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'P' $O/bench_driver.json
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d["roofline"]; print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"],4), "frac", round(r["frac"],4), "sustained_frac", r.get("sustained_frac"), "traffic", r.get("traffic"), "cpu", d.get("cpu_baseline",{}).get("value"))
P
timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4.json 2> $O/cfg4.err; echo cfg4 $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*' $O/cfg4.json | tr '\n' ' ')
timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 3 > $O/cfg3.json 2> $O/cfg3.err; echo cfg3 $(grep -o '"us_per_step": [0-9.]*\|"frac": [0-9.]*' $O/cfg3.json | tr '\n' ' ')
for n in 2 4; do
  PBL_BENCH_BACKEND=gloo PBL_BENCH_BASELINE=1 MASTER_PORT=296$n timeout 400 python bench.py --gpus $n --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline > $O/tp${n}_plumbing.json 2> $O/tp${n}_plumbing.err
  echo tp$n rc=$? $(grep -o '"tp_path": "[a-z0-9+-]*"\|"tp_notes": \[[^]]*\]' $O/tp${n}_plumbing.json | tr '\n' ' ')
done
