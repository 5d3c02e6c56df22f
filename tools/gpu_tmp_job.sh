VARIANTS='r2 r2abl1 r2abl5' bash tools/gpu_job6.sh xfabl r6n
O=gpurun_out/r6n
timeout 300 python bench.py --workload cfg3 --synth device --steps 10 --warmup 3 > $O/cfg3_xf.json 2> $O/cfg3_xf.err; python tools/show_line.py $O/cfg3_xf.json | cut -c1-600
PBL_GEMM_X_FRAGMENTS=0 timeout 300 python bench.py --workload cfg3 --synth device --steps 10 --warmup 3 > $O/cfg3_lds.json 2> $O/cfg3_lds.err; python tools/show_line.py $O/cfg3_lds.json | cut -c1-600
MODES=prefill BF16=0 timeout 600 python tools/bench_llama7b.py > $O/llama7b_xf.json 2> $O/llama7b_xf.err; tail -3 $O/llama7b_xf.json | cut -c1-800
