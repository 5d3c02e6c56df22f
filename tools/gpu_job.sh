#!/bin/bash
# What the gpurun calls of round 5 ran, by mode (each call: `gpurun -- 'bash tools/gpu_job.sh <mode> <tag>'`; output -> gpurun_out/<tag>/):
#   validate   full GPU suite (twice: flakiness), smoke, the driver's bench command, cfg3 / cfg4 with library defaults, 2- and 4-rank
#              plumbing lines of the multi-GPU default path on one device                                     (calls r5-1, r5-2, r5-6)
#   gemm       the GEMM regime per shape: llama-13b and gate/up at 2048 rows, llama-7b shapes at 300 / 512 / 1024 rows, cfg3, the
#              7B-shaped forward fp16 + bf16                                                                   (calls r5-1, r5-3, r5-4)
#   profiles   rocprofv3 kernel trace + PMC passes of the driver's command, of cfg4 and a kernel trace of cfg3            (call r5-3)
#   variants   A/B builds from tools/build_variant.sh (build/libpbl_<name>.so, PBL_LIB): cfg4 per variant                 (call r5-5)
#   gemvknobs  A/B builds of the headline GEMV's compile-time knobs (-DPBL_TILE_RING, -DPBL_GROUPED_WPB), the driver's command   (call r5k)
#   sbsplit    A/B builds of the small-batch kernel's K-split rule (-DPBL_SB_MIN_HPS) on llama-7b shapes, tools/bench_small.py   (call r5z)
#   bf16trace  kernel trace of a small decode batch with bf16 activations (tools/trace_bf16_small.py)                      (call r5x)
#   plumbing8  eight ranks of `bench.py --gpus 8` time-slicing one device (PBL_BENCH_BACKEND=gloo)                        (call r5-1)
set -u
export TMPDIR=/tmp
MODE=${1:-validate}; O=gpurun_out/${2:-r5}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
line() { python - "$1" <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d["roofline"]
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in dict(value=d["value"], ms_per_step=d["ms_per_step"], frac=r["frac"], sustained_frac=r.get("sustained_frac"),
              us_per_layer=r.get("us_per_layer"), us_per_step=r.get("us_per_step"), tp_path=d["config"].get("tp_path"), tp_notes=d["config"].get("tp_notes")).items() if v is not None})
P
}
case $MODE in
validate)
  for i in $(seq 1 ${REPS:-2}); do timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest$i.txt 2>&1; tail -3 $O/pytest$i.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest$i.txt | cut -c1-300; done
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; line $O/bench_driver.json
  timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4.json 2> $O/cfg4.err; line $O/cfg4.json
  timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 3 > $O/cfg3.json 2> $O/cfg3.err; line $O/cfg3.json
  timeout 200 python tools/bench_host.py > $O/host.json 2> $O/host.err; cut -c1-700 $O/host.json
  for n in 2 4; do
    PBL_BENCH_BACKEND=gloo PBL_BENCH_BASELINE=1 MASTER_PORT=296$n timeout 400 python bench.py --gpus $n --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline > $O/tp${n}_plumbing.json 2> $O/tp${n}_plumbing.err
    echo tp$n rc=$?; line $O/tp${n}_plumbing.json
  done ;;
plumbing8)
  PBL_BENCH_BACKEND=gloo PBL_BENCH_BASELINE=1 MASTER_PORT=2968 timeout 500 python bench.py --gpus 8 --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline > $O/tp8_plumbing.json 2> $O/tp8_plumbing.err
  echo tp8 rc=$?; line $O/tp8_plumbing.json ;;
gemm)
  PBL_BENCH_SHAPES=5120x5120:0.95,5120x13824:0.95,13824x5120:0.95,11008x4096:0.95 timeout 600 python tools/bench_gemm.py > $O/gemm_13b.jsonl 2> $O/gemm_13b.err; cut -c1-420 $O/gemm_13b.jsonl
  for m in 300 512 1024; do
    PBL_BENCH_M=$m PBL_BENCH_SHAPES=4096x4096:0.95,4096x11008:0.95,11008x4096:0.95 timeout 400 python tools/bench_gemm.py > $O/gemm_m$m.jsonl 2> $O/gemm_m$m.err; cut -c1-420 $O/gemm_m$m.jsonl
  done
  timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 3 > $O/cfg3.json 2> $O/cfg3.err; line $O/cfg3.json
  timeout 900 python tools/bench_llama7b.py > $O/llama7b.json 2> $O/llama7b.err; tail -1 $O/llama7b.json | cut -c1-1500 ;;
profiles)
  PROF_STEPS=20 PROF_WARMUP=5 timeout 900 bash tools/profile.sh r05_final > $O/profile_final.txt 2>&1; tail -30 $O/profile_final.txt | cut -c1-400
  timeout 600 bash tools/profile_cfg4.sh r05_cfg4 > $O/profile_cfg4.txt 2>&1; tail -12 $O/profile_cfg4.txt | cut -c1-400
  mkdir -p gpurun_out/prof_r05_cfg3
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05_cfg3/gemmimg_trace -o trace -- python bench.py --workload cfg3 --steps 10 --warmup 3 > gpurun_out/prof_r05_cfg3/trace.log 2>&1
  python tools/summarize_prof.py gpurun_out/prof_r05_cfg3 > gpurun_out/prof_r05_cfg3/summary.txt 2>&1; cut -c1-300 gpurun_out/prof_r05_cfg3/summary.txt | head -20
  find gpurun_out/prof_r05_final gpurun_out/prof_r05_cfg4 gpurun_out/prof_r05_cfg3 -name "*.db" -size +6M -delete ;;
variants)
  for v in default ${VARIANTS:-sbw5 sbw6 sbw7}; do
    if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
    timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4_$v.json 2> $O/cfg4_$v.err; echo cfg4 $v; line $O/cfg4_$v.json
  done ;;
gemvknobs)
  for v in default ${VARIANTS:-ring3 wpb2 wpb8} default; do
    if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "== $v"; line $O/bench_$v.json
  done ;;
sbsplit)
  for v in default ${VARIANTS:-hps2 hps3 hps6}; do
    if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
    PBL_BENCH_SHAPES=${SHAPES:-4096x4096:0.95,11008x4096:0.95,4096x11008:0.95,4096x4096:0.9} PBL_BENCH_MS=${MS:-32,16} timeout 300 python tools/bench_small.py > $O/small_$v.jsonl 2> $O/small_$v.err
    echo "== $v"; python -c "
import json,sys
for l in open('$O/small_$v.jsonl'):
    d=json.loads(l); print(d['shape'], d['low_frac'], d['M'], 'image', d.get('image_us_w2048'), 'records', d['records_us'], 'dense', d['dense_us'])"
  done ;;
bf16trace)
  mkdir -p gpurun_out/prof_r05_bf16
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05_bf16/trace -o trace -- python tools/trace_bf16_small.py > gpurun_out/prof_r05_bf16/trace.log 2>&1
  tail -2 gpurun_out/prof_r05_bf16/trace.log | cut -c1-400
  python tools/summarize_prof.py gpurun_out/prof_r05_bf16 > gpurun_out/prof_r05_bf16/summary.txt 2>&1; cut -c1-300 gpurun_out/prof_r05_bf16/summary.txt | head -24 ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
