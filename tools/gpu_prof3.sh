set -u
export TMPDIR=/tmp
O=gpurun_out/prof_r02e
mkdir -p $O
timeout 600 python -m pytest -m gpu -q tests/test_gpu_parity.py -k "unpack or weight or dense" tests/test_gpu_fuzz.py > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
export PBL_BENCH_CACHE=/tmp/mfma_cache3.pt
python tools/bench_mfma.py > $O/mfma_plain.log 2>&1; tail -1 $O/mfma_plain.log
python tools/bench_unpack.py > $O/unpack_plain.log 2>&1; tail -1 $O/unpack_plain.log
for t in unpack qat ptq pack gemm mfma; do
  rocprofv3 --kernel-trace --stats -d $O/${t}_trace -o trace -- python tools/bench_$t.py > $O/${t}_trace.log 2>&1
  tail -1 $O/${t}_trace.log | cut -c1-300
done
python tools/summarize_prof.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
du -sh $O
