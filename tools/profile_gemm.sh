set -u
export TMPDIR=/tmp
O=gpurun_out/prof_r05_gemm; mkdir -p $O
export PBL_BENCH_SHAPES=4096x4096:0.95,5120x5120:0.95 PBL_BENCH_ONLY=fused
rocprofv3 --kernel-trace --stats -d $O/gemmimg_trace -o trace -- python tools/bench_gemm.py > $O/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/gemmimg_pmc -o pmc -- python tools/bench_gemm.py > $O/pmc.log 2>&1
python tools/summarize_prof.py $O > $O/summary.txt 2>&1
find $O -name "*.db" -size +6M -delete
cut -c1-330 $O/summary.txt | head -30; tail -2 $O/trace.log | cut -c1-400
