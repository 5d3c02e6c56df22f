#!/bin/bash
# One gpurun call (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# r4-3: component builds of the GEMM-regime kernel under the timeline probe: real clock + span of each (is the MFMA pipe full?)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r43}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
for v in t_alt1 a_mfr a_bare a_noexp a_nox a_nomfma a_nofrag t_alt1; do
  PBL_LIB=build/libpbl_$v.so timeout 100 python tools/trace_gemm.py 4096x4096:0.95 >> $O/trace.jsonl 2>> $O/trace.err
done
python - <<'P' $O/trace.jsonl
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["lib"], d["us_per_call_events"], "span", d["kernel_span_us"], "clk", d["sclk_mhz"], "cons start/loop/tail", d["consumer"]["startup_us"]["p50"], d["consumer"]["loop_us"]["p50"], d["consumer"]["tail_us"]["p50"],
          "bar by wave", d["barrier_wait_by_wave_us"], "vm", d["vmcnt_wait_by_wave_us"])
P
