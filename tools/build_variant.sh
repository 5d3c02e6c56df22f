#!/bin/bash
# build an A/B variant of libpbl.so with extra -D flags: tools/build_variant.sh <name> -DFOO=1 ...  -> build/libpbl_<name>.so
set -e
name=$1; shift
mkdir -p build
C=pb_llm_amd/csrc
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -shared --offload-arch=gfx950 -Wno-unused-function "$@" \
  $C/pbl_kernels.hip $C/pbl_gemm.hip $C/pbl_gemm_big.hip $C/pbl_gemm_img.hip $C/pbl_act.hip $C/pbl_qat.hip $C/pbl_prep.hip $C/pbl_comm.hip $C/pbl_pack.hip $C/pbl_host.cpp -o build/libpbl_$name.so
echo build/libpbl_$name.so
