#!/bin/bash
# r4-19: stream-K timeline
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r419}; mkdir -p $O
PBL_LIB=build/libpbl_trace.so timeout 600 python tools/trace_sk.py 11008x4096:0.95 2>&1 | tee $O/trace_sk.jsonl | cut -c1-1500
