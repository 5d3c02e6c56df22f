#!/bin/bash
# gpurun jobs of round 6, by mode (each call: `gpurun -- 'bash tools/gpu_job6.sh <mode> <tag>'`; output -> gpurun_out/<tag>/)
set -u
export TMPDIR=/tmp
MODE=${1:-quick}; O=gpurun_out/${2:-r6}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
case $MODE in
bench)   # the driver's command (with the side entries)
  T0=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench wall $(echo "$(date +%s.%N) - $T0" | bc) s"; tail -2 $O/bench_driver.err | cut -c1-300
  python tools/show_line.py $O/bench_driver.json
  ;;
quick)   # the round's new tests, smoke, the driver's command (with the side entries)
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -k "${KEXPR:-full_tensor or beyond_fp16 or split_and_join}" > $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest_new.txt | cut -c1-300
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
  T0=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench wall $(echo "$(date +%s.%N) - $T0" | bc) s"; tail -2 $O/bench_driver.err | cut -c1-300
  python tools/show_line.py $O/bench_driver.json
  ;;
small)   # the small-batch kernel's geometries: parity tests, then a sweep of forced plans per shape (tools/bench_small.py)
  timeout 1500 python -m pytest tests/test_gpu_gemm.py -q -m gpu -p no:cacheprovider --timeout 900 -k "small" > $O/pytest_small.txt 2>&1; tail -3 $O/pytest_small.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest_small.txt | cut -c1-300
  PBL_BENCH_SHAPES=${SHAPES:-13824x5120:0.8,5120x13824:0.8} PBL_BENCH_MS=${MS:-32} PBL_SB_PLANS=${PLANS:-default,0:0,1:1,1:2,2:1,2:2,2:3,2:4,2:6} timeout 900 python tools/bench_small.py > $O/small.jsonl 2> $O/small.err
  python - $O/small.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d['shape'], d['low_frac'], 'M', d['M'], 'img_MB', d['image_MB'], 'blob_MB', d['blob_MB'], {k[5:-3]: v for k, v in d.items() if k.startswith('plan_') and k.endswith('_us')}, 'records', d['records_us'], 'dense', d['dense_us'], 'maxerr', max(v for k, v in d.items() if k.endswith('_err')))
P
  tail -3 $O/small.err | cut -c1-300
  ;;
smallvar)   # A/B builds of the small-batch kernel (tools/build_variant.sh -> build/libpbl_<name>.so) on the cfg4 shapes + the per-CU bandwidth probe
  [ -x build/ubench_cu_bw ] && timeout 120 build/ubench_cu_bw > $O/cu_bw.json 2>&1; python - $O/cu_bw.json <<'P'
import json,sys
try:
    for r in json.load(open(sys.argv[1]))["rows"]:
        if r: print(r)
except Exception as e: print("cu_bw", e)
P
  for v in default ${VARIANTS:-d4 nt d4nt}; do
    if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
    PBL_BENCH_SHAPES=${SHAPES:-13824x5120:0.8,5120x13824:0.8} PBL_BENCH_MS=${MS:-32} PBL_SB_PLANS=${PLANS:-default,0:0,1:1} timeout 600 python tools/bench_small.py > $O/small_$v.jsonl 2> $O/small_$v.err
    echo "== $v"; python - $O/small_$v.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d['shape'], 'M', d['M'], {k[5:-3]: v for k, v in d.items() if k.startswith('plan_') and k.endswith('_us')}, 'maxerr', max(v for k, v in d.items() if k.endswith('_err')))
P
  done
  ;;
mix1)   # image-only residency test + new small-batch tests, GEMM-image kernel ablations (A/B builds), hessian-vs-magnitude GEMV
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -k "image_only or small_batch or fused_decode or merged or fp32_act" > $O/pytest_mix.txt 2>&1; tail -3 $O/pytest_mix.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest_mix.txt | cut -c1-300
  for v in default abl1 abl2 abl3 abl4 abl6 abl7 default; do
    if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
    [ $v = default ] || [ -f build/libpbl_$v.so ] || continue
    timeout 200 python tools/bench_gemm_ablate.py 2>> $O/ablate.err | tee -a $O/ablate.jsonl
  done
  unset PBL_LIB
  timeout 300 python tools/bench_hessian_gemv.py 2> $O/hess.err | tee $O/hessian_gemv.jsonl | cut -c1-400
  ;;
profiles)   # rocprofv3 of the driver's command (kernel trace + PMC passes), of cfg4 (trace + PMC) and a kernel trace of cfg3 -- the side entries' layers
  [ -x build/calib_fetch ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/calib_fetch.hip -o build/calib_fetch > /dev/null 2>&1
  PROF_STEPS=20 PROF_WARMUP=5 timeout 1200 bash tools/profile.sh r06_final > $O/profile_final.txt 2>&1; tail -30 $O/profile_final.txt | cut -c1-400
  timeout 900 bash tools/profile_cfg4.sh r06_cfg4 --synth device > $O/profile_cfg4.txt 2>&1; tail -14 $O/profile_cfg4.txt | cut -c1-400
  mkdir -p gpurun_out/prof_r06_cfg3
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06_cfg3/gemmimg_trace -o trace -- python bench.py --workload cfg3 --synth device --steps 10 --warmup 3 > gpurun_out/prof_r06_cfg3/trace.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof_r06_cfg3/gemmimg_pmc -o pmc -- python bench.py --workload cfg3 --synth device --steps 10 --warmup 3 > gpurun_out/prof_r06_cfg3/pmc.log 2>&1
  python tools/summarize_prof.py gpurun_out/prof_r06_cfg3 > gpurun_out/prof_r06_cfg3/summary.txt 2>&1; cut -c1-300 gpurun_out/prof_r06_cfg3/summary.txt | head -24
  grep '"metric"' gpurun_out/prof_r06_cfg3/trace.log | cut -c1-600
  find gpurun_out/prof_r06_final gpurun_out/prof_r06_cfg4 gpurun_out/prof_r06_cfg3 -name "*.db" -size +6M -delete ;;
mix2)   # x-staging byte-reduction ablations of the GEMM-image kernel; tensor-parallel plumbing lines with the phase timings (ranks share ONE GPU)
  for v in default abl4 abl12 abl20 abl8 abl16 default; do
    if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
    [ $v = default ] || [ -f build/libpbl_$v.so ] || continue
    timeout 200 python tools/bench_gemm_ablate.py 2>> $O/ablate.err | tee -a $O/ablate2.jsonl
  done
  unset PBL_LIB
  for n in 2 4; do
    PBL_BENCH_BACKEND=gloo PBL_BENCH_BASELINE=1 MASTER_PORT=296$n timeout 500 python bench.py --gpus $n --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline > $O/tp${n}_plumbing.json 2> $O/tp${n}_plumbing.err
    echo tp$n rc=$?; python - $O/tp${n}_plumbing.json <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(round(d["value"]), d["config"].get("tp_path"), d["config"].get("tp_notes"), json.dumps(d.get("tp_phases"))[:900])
P
    tail -2 $O/tp${n}_plumbing.err | cut -c1-300
  done
  ;;
xf)   # x as a fragment-major copy: parity (bit-identical to the LDS-staged kernel) + timing per shape
  timeout 1200 python -m pytest tests/test_gpu_gemm.py -q -m gpu -p no:cacheprovider --timeout 900 -k "fragment_major or full_tensor" > $O/pytest_xf.txt 2>&1; tail -3 $O/pytest_xf.txt | cut -c1-300; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_xf.txt | head -20 | cut -c1-300
  for shp in ${SHAPES:-4096x4096 11008x4096 4096x11008 5120x5120}; do
    PBL_BENCH_SHAPE=$shp timeout 300 python tools/bench_gemm_ablate.py 2>> $O/xf.err | tee -a $O/xf.jsonl
  done
  ;;
xfabl)   # the xf tests + the GEMM-image kernel (both x paths) in A/B builds with parts of the loop removed
  timeout 1200 python -m pytest tests/test_gpu_gemm.py -q -m gpu -p no:cacheprovider --timeout 900 -k "fragment_major or full_tensor" > $O/pytest_xf.txt 2>&1; tail -3 $O/pytest_xf.txt | cut -c1-300; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_xf.txt | head -20 | cut -c1-300
  for v in default ${VARIANTS:-abl1 abl4 abl5} default; do
    if [ $v = default ]; then unset PBL_LIB; else export PBL_LIB=build/libpbl_$v.so; fi
    [ $v = default ] || [ -f build/libpbl_$v.so ] || continue
    for shp in ${SHAPES:-4096x4096}; do
      PBL_BENCH_SHAPE=$shp timeout 300 python tools/bench_gemm_ablate.py 2>> $O/xfabl.err | tee -a $O/xfabl.jsonl
    done
  done
  unset PBL_LIB
  ;;
profcfg3)   # rocprofv3 of the cfg3 side workload alone (kernel trace + one SQ counter pass) + the driver's command once
  mkdir -p gpurun_out/prof_r06_cfg3
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06_cfg3/gemmimg_trace -o trace -- python bench.py --workload cfg3 --synth device --steps 10 --warmup 3 > gpurun_out/prof_r06_cfg3/trace.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof_r06_cfg3/gemmimg_pmc -o pmc -- python bench.py --workload cfg3 --synth device --steps 10 --warmup 3 > gpurun_out/prof_r06_cfg3/pmc.log 2>&1
  python tools/summarize_prof.py gpurun_out/prof_r06_cfg3 > gpurun_out/prof_r06_cfg3/summary.txt 2>&1; cut -c1-300 gpurun_out/prof_r06_cfg3/summary.txt | head -24
  grep '"metric"' gpurun_out/prof_r06_cfg3/trace.log | cut -c1-600
  find gpurun_out/prof_r06_cfg3 -name "*.db" -size +6M -delete
  T0=$(date +%s.%N); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench wall $(echo "$(date +%s.%N) - $T0" | bc) s"
  python tools/show_line.py $O/bench_driver.json
  ;;
full)
  for i in $(seq 1 ${REPS:-1}); do timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest$i.txt 2>&1; tail -3 $O/pytest$i.txt | cut -c1-300; grep -E "^(FAILED|ERROR)" $O/pytest$i.txt | cut -c1-300; done
  ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
