#!/bin/bash
# One gpurun call of round 5 (the validation batch: full GPU suite, smoke, the driver's bench command, the side workloads, the
# llama-13b GEMM shapes, the multi-rank plumbing lines on one device, the host cost of a bf16 call).  Usage: tools/gpu_job.sh <tag>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r5}; mkdir -p $O
timeout 300 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt | cut -c1-400
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'P' $O/bench_driver.json
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d["roofline"]; print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"],4), "frac", round(r["frac"],4), "sustained_frac", r.get("sustained_frac"), "cpu", d.get("cpu_baseline",{}).get("value"))
P
timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 > $O/cfg4.json 2> $O/cfg4.err; echo cfg4 $(grep -o '"us_per_layer": [0-9.]*\|"frac": [0-9.]*\|"small_batch_image": "[a-z0-9]*"' $O/cfg4.json | tr '\n' ' ')
timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 3 > $O/cfg3.json 2> $O/cfg3.err; echo cfg3 $(grep -o '"us_per_step": [0-9.]*\|"frac": [0-9.]*\|"gemm_backend": "[a-z]*"' $O/cfg3.json | tr '\n' ' ')
timeout 300 python bench.py --workload cfg3 --gemm-backend tuned --steps 10 --warmup 3 > $O/cfg3_tuned.json 2> $O/cfg3_tuned.err; echo cfg3_tuned $(grep -o '"us_per_step": [0-9.]*\|"frac": [0-9.]*' $O/cfg3_tuned.json | tr '\n' ' ')
PBL_BENCH_SHAPES=5120x5120:0.95,13824x5120:0.95,5120x13824:0.95,11008x4096:0.95 timeout 600 python tools/bench_gemm.py > $O/gemm_13b.jsonl 2> $O/gemm_13b.err; cut -c1-330 $O/gemm_13b.jsonl
for n in 2 8; do
  PBL_BENCH_BACKEND=gloo PBL_BENCH_BASELINE=1 MASTER_PORT=295$n timeout 500 python bench.py --gpus $n --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline > $O/tp${n}_plumbing.json 2> $O/tp${n}_plumbing.err
  echo tp$n rc=$? $(grep -o '"tp_path": "[a-z0-9+-]*"\|"ms_per_step": [0-9.]*\|"tp_notes": \[[^]]*\]' $O/tp${n}_plumbing.json | tr '\n' ' '); tail -3 $O/tp${n}_plumbing.err | cut -c1-300
done
timeout 300 python tools/bench_host.py > $O/host.json 2> $O/host.err; cut -c1-900 $O/host.json
