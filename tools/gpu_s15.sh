set -u
export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O
timeout 900 python tools/bench_prefetch.py --gap 4 > $O/prefetch_gap4.json 2>$O/err4.log; tail -1 $O/prefetch_gap4.json; tail -3 $O/err4.log
timeout 900 python tools/bench_prefetch.py --gap 10 > $O/prefetch_gap10.json 2>$O/err10.log; tail -1 $O/prefetch_gap10.json
