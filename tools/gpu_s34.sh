set -u
export TMPDIR=/tmp
O=gpurun_out/s34; mkdir -p $O
for v in libpbl_proto_r1.so libpbl_proto_r4.so libpbl_proto_r8.so; do
  PBL_PROTO_LIB=$v timeout 900 python tools/proto_rowmajor.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k: (v['us_per_launch_200'], v.get('max_abs_diff_vs_shipped')) for k,v in d.items() if isinstance(v, dict)})"
done | tee $O/proto_rpw.txt
