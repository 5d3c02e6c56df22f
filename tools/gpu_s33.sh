set -u
export TMPDIR=/tmp
O=gpurun_out/s33; mkdir -p $O
for v in base p1 p4 p13; do
  if [ $v = base ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
  for i in 1 2; do python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"; done
done | tee $O/prio.txt
