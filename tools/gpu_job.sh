#!/bin/bash
# r4-35: reduce kernel with one round of loads; slot loads with / without the non-temporal hint (cfg4 bench + kernel trace)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r435}; mkdir -p $O
for lib in pb_llm_amd/libpbl.so build/libpbl_sbnt.so; do
  n=$(basename $lib .so)
  PBL_LIB=$lib PBL_NATIVE=0 timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4_$n.json 2> $O/cfg4_$n.err; echo $n ctypes-route $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*' $O/cfg4_$n.json | tr '\n' ' ')
done
timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4.json 2> $O/cfg4.err; echo native $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*' $O/cfg4.json | tr '\n' ' ')
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/trace.log 2>&1
python - <<'P' $O/trace
import sys, glob, sqlite3, collections
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor(); agg = collections.defaultdict(list)
    for n, d in cur.execute("select name, duration from kernels"): agg[n].append(d)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:3]: print(k[:60], len(v), round(sum(v)/len(v)/1e3, 2))
P
find $O -name "*.db" -size +8M -delete
