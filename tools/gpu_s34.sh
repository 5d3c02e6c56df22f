set -u
export TMPDIR=/tmp
O=gpurun_out/s34; mkdir -p $O
timeout 900 python tools/proto_rowmajor.py 2>/dev/null | tail -1 | tee $O/proto.json | cut -c1-900
