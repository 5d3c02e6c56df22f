set -u
export TMPDIR=/tmp
O=gpurun_out/s19; mkdir -p $O
timeout 900 python -m pytest -m gpu -q tests/test_gpu_configs.py -k "k_split or routing or zero_valued or mfma" > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log; tail -15 $O/t1.log
timeout 900 python -m pytest -m gpu -q tests/test_gpu_groups.py tests/test_gpu_fuzz.py > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log; tail -3 $O/t2.log
PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9" PBL_BENCH_M=32,16 timeout 600 python tools/bench_mfma.py 2>&1 | tail -1 | tee $O/mfma.json
PBL_BENCH_M=1,2,3,4,5,8 timeout 1200 python tools/bench_route.py 2>&1 | tee $O/route.txt | tail -4
