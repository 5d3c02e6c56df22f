#!/bin/bash
# r4-17: the driver's bench command (with the sustained figure), its rocprofv3 profile split into pre-heat / timed window,
# traffic passes, cfg4 side workload
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r417}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'P' $O/bench_driver.json
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d["roofline"]; print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"],4), "frac", round(r["frac"],4), "sustained_frac", r.get("sustained_frac"), "sustained_us", r.get("sustained_us_per_launch"), "n", r.get("sustained_launches"), "cpu", d.get("cpu_baseline",{}).get("value"))
P
PROF_STEPS=20 PROF_WARMUP=5 timeout 900 bash tools/profile.sh r04 > $O/profile.log 2>&1; tail -40 gpurun_out/prof_r04/summary.txt | cut -c1-600
find gpurun_out/prof_r04 -name "*.db" -size +8M -delete
timeout 200 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4.json 2> $O/cfg4.err; cut -c1-300 $O/cfg4.json; grep -o '"us_per_layer": [0-9.]*' $O/cfg4.json
