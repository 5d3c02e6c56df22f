#!/bin/bash
# Ablated variants of the MFMA small-batch kernel (performance analysis only; results are wrong by design).
#   tools/ablate_mfma.sh build "0 2 64 ..."   here (no GPU): builds pb_llm_amd/_ablate/libpbl_<A>.so
#   tools/ablate_mfma.sh run                  on the GPU box: times every prebuilt variant
set -u
SRC="pb_llm_amd/csrc/pbl_kernels.hip pb_llm_amd/csrc/pbl_gemm.hip pb_llm_amd/csrc/pbl_qat.hip pb_llm_amd/csrc/pbl_prep.hip pb_llm_amd/csrc/pbl_host.cpp"
D=pb_llm_amd/_ablate
if [ "${1:-run}" = build ]; then
  rm -rf $D; mkdir -p $D
  for A in ${2:-0 2 4 8 16}; do
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -shared --offload-arch=gfx950 -Iinclude -DPBL_MFMA_ABLATE=$A $SRC -o $D/libpbl_$A.so 2>/dev/null &
  done
  wait; ls $D
else
  cp pb_llm_amd/libpbl.so /tmp/libpbl_orig.so
  for f in $(ls $D/libpbl_*.so | sort -t_ -k3 -n); do
    cp $f pb_llm_amd/libpbl.so
    echo "== $(basename $f)  (bits: 1 no sort, 2 no scatter, 4 no mfma, 8 no clears, 16 no x loads, 64 no scatter writes, 128 const A frags, 512 sort only)"
    python tools/bench_mfma.py 2>&1 | tail -1
  done
  cp /tmp/libpbl_orig.so pb_llm_amd/libpbl.so
fi
