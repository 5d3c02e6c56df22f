set -u
export TMPDIR=/tmp
O=gpurun_out/prof_r02b
mkdir -p $O
export PBL_BENCH_SHAPES="4096x4096:0.95" PBL_BENCH_ONLY=fused
rocprofv3 --kernel-trace --stats -d $O/gemm_trace -o trace -- python tools/bench_gemm.py > $O/gemm_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $O/gemm_pmc1 -o pmc -- python tools/bench_gemm.py > $O/gemm_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace -d $O/gemm_pmc2 -o pmc -- python tools/bench_gemm.py > $O/gemm_pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace -d $O/gemm_pmc3 -o pmc -- python tools/bench_gemm.py > $O/gemm_pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/gemm_fetch -o pmc -- python tools/bench_gemm.py > $O/gemm_fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/gemm_tcc -o pmc -- python tools/bench_gemm.py > $O/gemm_tcc.log 2>&1
python tools/summarize_prof.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +3M -delete
find $O -name "*.db" -delete
cat $O/summary.txt
