set -u
export TMPDIR=/tmp
O=gpurun_out/prof_r02f
mkdir -p $O
# A. the driver's command under the kernel trace (final kernels of the round)
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/trace.log 2>&1
tail -c 600 $O/trace.log
# B. matrix-core kernel on config 4 (13824x5120, low_frac 0.8), M = 32, after the scatter rewrite
export PBL_BENCH_SHAPES="13824x5120:0.8" PBL_BENCH_M=32 PBL_BENCH_CACHE=/tmp/mfma_cache.pt
python tools/bench_mfma.py > $O/mfma_plain.log 2>&1; tail -1 $O/mfma_plain.log
rocprofv3 --kernel-trace --stats -d $O/mfma_trace -o trace -- python tools/bench_mfma.py > $O/mfma_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $O/mfma_pmc1 -o pmc -- python tools/bench_mfma.py > $O/mfma_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace -d $O/mfma_pmc2 -o pmc -- python tools/bench_mfma.py > $O/mfma_pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace -d $O/mfma_pmc3 -o pmc -- python tools/bench_mfma.py > $O/mfma_pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/mfma_fetch -o pmc -- python tools/bench_mfma.py > $O/mfma_fetch.log 2>&1
python tools/summarize_prof.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
cat $O/summary.txt | cut -c1-400
