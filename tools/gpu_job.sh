#!/bin/bash
# r4-41: final image layout: llama-7b-shaped prefill, kernel trace + counters of the image GEMM kernel at 4096^2 x 2048
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r441}; mkdir -p $O
MODES=prefill timeout 600 python tools/bench_llama7b.py 2>&1 | grep -v amdgpu.ids | tee $O/llama7b.jsonl | cut -c1-700
P=gpurun_out/prof_r04g; mkdir -p $P
CMD="python tools/bench_gemm.py"
export PBL_BENCH_SHAPES=4096x4096:0.95 PBL_BENCH_ONLY=fused
rocprofv3 --kernel-trace --stats -d $P/gemmimg_trace -o trace -- $CMD > $P/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $P/gemmimg_pmc -o pmc -- $CMD > $P/pmc.log 2>&1
python tools/summarize_prof.py $P 2>&1 | grep -A3 "gemmimg" | cut -c1-700
find $P -name "*.db" -size +8M -delete
