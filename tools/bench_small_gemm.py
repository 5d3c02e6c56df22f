#!/usr/bin/env python3
"""Time fp16-checkpoint layers (PBL_FLAG_SAL_F16) at small batch: band GEMM vs GEMV passes vs dense path."""
import json, sys, os
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q

def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for shp, lf in (("13824x5120", 0.8), ("5120x13824", 0.8), ("4096x4096", 0.9)):
    N, K = map(int, shp.split("x"))
    W = synth.llm_weight(N, K, seed=N % 97)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    ncopy = max(2, int(0.6e9 / (N * K * 0.3)))
    base = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])
    layers = [Q.PBLinear(base.packed.to("cuda:0"), None) for _ in range(ncopy)]
    for M in (16, 32, 64):
        x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
        res = {}
        for name, lo, hi in (("band_gemm", 12, 64), ("gemv_passes", 10 ** 6, 64), ("dense_lib_gemm", 12, 0)):
            Q.GEMM_THRESHOLD, Q.SMALL_GEMM_MAX, Q.SMALL_GEMM_MIN_RECORDS = lo, hi, 1
            def run():
                for l in layers: l(x)
            res[name] = round(timeit(run, 3) / ncopy, 1)
        Q.GEMM_THRESHOLD, Q.SMALL_GEMM_MAX = 12, 64
        print(json.dumps(dict(shape=shp, low_frac=lf, M=M, us_per_call=res, gflops_band=round(2.0 * N * K * M / res["band_gemm"] / 1e3))), flush=True)
