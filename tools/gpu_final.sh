set -u
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest -m gpu -q tests > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log; tail -6 $O/full.log
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; tail -1 $O/bench.json | cut -c1-1500
