#!/bin/bash
# FETCH_SIZE of the grouped GEMV at 4 vs 8 waves per workgroup (x staged once per workgroup)
export TMPDIR=/tmp
SRC="pb_llm_amd/csrc/pbl_kernels.hip pb_llm_amd/csrc/pbl_gemm.hip pb_llm_amd/csrc/pbl_qat.hip pb_llm_amd/csrc/pbl_prep.hip pb_llm_amd/csrc/pbl_host.cpp"
cp pb_llm_amd/libpbl.so /tmp/libpbl_orig.so
for W in 4 8; do
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -shared --offload-arch=gfx950 -DPBL_GROUPED_WPB=$W $SRC -o pb_llm_amd/libpbl.so 2>/dev/null
  touch pb_llm_amd/libpbl.so
  rm -rf gpurun_out/fw$W
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/fw$W -o pmc -- python bench.py --steps 60 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
db=glob.glob("gpurun_out/fw$W/**/*.db", recursive=True)[0]
cur=sqlite3.connect(db).cursor()
v=[r[0] for r in cur.execute("select value from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%pbl_gemv%'")]
print("WPB=$W FETCH_SIZE KiB avg", sum(v)/len(v), "-> x2 MB", sum(v)/len(v)*2*1024/1e6)
PY
done
cp /tmp/libpbl_orig.so pb_llm_amd/libpbl.so
