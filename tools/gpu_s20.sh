set -u
export TMPDIR=/tmp
O=gpurun_out/s20; mkdir -p $O
timeout 1500 python -m pytest -m gpu -q tests > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log; tail -8 $O/full.log
PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9" PBL_BENCH_M=32,16 timeout 600 python tools/bench_mfma.py 2>&1 | tail -1 | tee $O/mfma.json
