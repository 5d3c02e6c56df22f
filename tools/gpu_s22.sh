set -u
export TMPDIR=/tmp
O=gpurun_out/s22; mkdir -p $O
export PBL_BENCH_CACHE=/tmp/c5.pt PBL_BENCH_SHAPES="4096x4096:0.95,11008x4096:0.95,4096x11008:0.95,13824x5120:0.8,5120x13824:0.8"
PBL_BENCH_M=16 python tools/bench_mfma.py > /dev/null 2>&1
for v in upf0 upf1; do PBL_LIB=build/libpbl_$v.so python tools/bench_unpack.py 2>&1 | tail -1 | sed "s/^/$v /"; done | tee $O/unpack_ab.txt
