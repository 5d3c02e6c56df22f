#!/bin/bash
# gpurun jobs of round 6, by mode (each call: `gpurun -- 'bash tools/gpu_job6.sh <mode> <tag>'`; output -> gpurun_out/<tag>/)
set -u
export TMPDIR=/tmp
MODE=${1:-quick}; O=gpurun_out/${2:-r6}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
case $MODE in
bench)   # the driver's command (with the side entries)
  T0=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench wall $(echo "$(date +%s.%N) - $T0" | bc) s"; tail -2 $O/bench_driver.err | cut -c1-300
  python tools/show_line.py $O/bench_driver.json
  ;;
quick)   # the round's new tests, smoke, the driver's command (with the side entries)
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -k "${KEXPR:-full_tensor or beyond_fp16 or split_and_join}" > $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest_new.txt | cut -c1-300
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
  T0=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench wall $(echo "$(date +%s.%N) - $T0" | bc) s"; tail -2 $O/bench_driver.err | cut -c1-300
  python tools/show_line.py $O/bench_driver.json
  ;;
full)
  for i in $(seq 1 ${REPS:-1}); do timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest$i.txt 2>&1; tail -3 $O/pytest$i.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest$i.txt | cut -c1-300; done
  ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
