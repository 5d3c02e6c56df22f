set -u
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
for v in rpw2t rpw4t; do
  PBL_LIB=build/libpbl_$v.so timeout 600 python -m pytest -m gpu -q tests/test_gpu_parity.py::test_grouped_launch_matches_individual tests/test_gpu_configs.py::test_fused_decode_and_graph_replay_bit_for_bit "tests/test_gpu_parity.py::test_hipgraph_capture_and_side_stream" > $O/test_$v.log 2>&1; echo "rc=$?" >> $O/test_$v.log; tail -3 $O/test_$v.log
done
for v in base rpw2 rpw4 rpw2n rpw4n; do
  if [ $v = base ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'])"; done
done 2>&1 | tee $O/bench.txt
