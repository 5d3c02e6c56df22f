set -u
export TMPDIR=/tmp
O=gpurun_out/s18; mkdir -p $O
PBL_BENCH_M=1,2,3,4 timeout 1200 python tools/bench_route.py 2>&1 | tee $O/route_small.txt | tail -6
