#!/bin/bash
# Sample GPU clocks / power in the middle of a long sustained bench run.
python bench.py --steps 120000 --warmup 300 --no-cpu-baseline > /tmp/b.log 2>&1 &
BP=$!
sleep 28
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|junction|Temperature \(Sensor hbm" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 2
done
wait $BP
tail -1 /tmp/b.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  layer-tok/s %.0f  GB/s %.0f  us/layer %.3f' % (j['value'], j['roofline']['achieved'], j['roofline']['us_per_layer']))"
rocm-smi --showmaxpower --showclocks 2>/dev/null | grep -E "Max|sclk" | tr -s ' \t' ' ' | head -4
