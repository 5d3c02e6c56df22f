#!/usr/bin/env python3
"""Time pbl_gemm_mfma_f16 alone on cached packed layers (tools/ablate_mfma.sh drives it)."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q
from pb_llm_amd.packing import PackedWeight

CACHE = os.environ.get("PBL_BENCH_CACHE", "/tmp/pbl_mfma_cache.pt")
# NxK:low_frac[:groupsize]
SHAPES = tuple((t, float(t.split(":")[1])) for t in os.environ.get(
    "PBL_BENCH_SHAPES", "13824x5120:0.8,5120x13824:0.8,4096x4096:0.9").split(","))
if os.path.exists(CACHE):
    blobs = torch.load(CACHE)
else:
    blobs = {}
    for shp, lf in SHAPES:
        N, K = map(int, shp.split(":")[0].split("x"))
        gs = int(shp.split(":")[2]) if shp.count(":") > 1 else -1
        W = synth.llm_weight(N, K, seed=N % 97)
        mask = O.ptq_low_mask(W, lf, "magnitude", None, gs)
        r = O.ptq_rtn(W, mask, 8, gs)
        base = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), gs, r["hscale"], r["hzero"])
        blobs[shp] = base.packed.blob.cpu()
    torch.save(blobs, CACHE)
out = {}
for shp, _ in SHAPES:
    N, K = map(int, shp.split(":")[0].split("x"))
    pk = PackedWeight.from_blob(blobs[shp])
    ncopy = max(2, int(float(os.environ.get("PBL_BENCH_BYTES", "0.6e9")) / (N * K * 0.3)))
    layers = [pk.to("cuda:0") for _ in range(ncopy)]
    for M in tuple(int(m) for m in os.environ.get("PBL_BENCH_M", "16,32").split(",")):
        x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
        def run():
            for l in layers: Q.mfma_forward(l, None, x)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        out[f"{shp}/M{M}"] = round(e0.elapsed_time(e1) * 1e3 / 5 / ncopy, 1)
print(json.dumps(out))
