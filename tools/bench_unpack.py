#!/usr/bin/env python3
"""pbl_unpack_dev alone: microseconds and output GB/s (the dense fp16 matrix is the algorithmic traffic)."""
import json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import PackedWeight
sys.argv = [sys.argv[0]]
CACHE = os.environ.get("PBL_BENCH_CACHE", "/tmp/pbl_mfma_cache.pt")
if not os.path.exists(CACHE):
    import runpy; runpy.run_path(os.path.join(REPO, "tools", "bench_mfma.py"))
blobs = torch.load(CACHE)
out = {}
for shp, blob in blobs.items():
    N, K = map(int, shp.split(":")[0].split("x"))
    pk = PackedWeight.from_blob(blob).to("cuda:0")
    for dt in (torch.float16, torch.float32):
        Q.unpack_on_device(pk, dt); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): Q.unpack_on_device(pk, dt)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out[f"{shp}/{'f16' if dt == torch.float16 else 'f32'}"] = dict(us=round(us, 1), out_GBps=round(N * K * (2 if dt == torch.float16 else 4) / us / 1e3))
print(json.dumps(out))
