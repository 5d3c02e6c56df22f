#!/usr/bin/env python3
"""Host packer (pbl_pack_dense_f32) wall time by thread count, one 4096x4096 layer (size query + fill)."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import pb_oracle as O
from pb_llm_amd import synth
from pb_llm_amd.packing import pack_dense
W = synth.llm_weight(4096, 4096, seed=1)
mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
r = O.ptq_rtn(W, mask, 8, -1)
hi, lo = r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0]
best = 1e9
for _ in range(3):
    t = time.time(); p = pack_dense(r["W_fq"], hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8)); best = min(best, time.time() - t)
print(best)
''' % REPO
out = {"host_cpus": os.cpu_count()}
for th in (1, 4, 16, 64):
    env = dict(os.environ, PBL_PACK_THREADS=str(th))
    out[f"threads_{th}_s"] = round(float(subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]), 3)
print(json.dumps(out))
