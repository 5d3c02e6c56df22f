#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r416}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1; tail -1 $O/build.txt
timeout 300 python tools/dbg_img.py > $O/dbg.txt 2>&1; tail -20 $O/dbg.txt
