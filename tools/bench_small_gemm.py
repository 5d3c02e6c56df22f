#!/usr/bin/env python3
"""Time fp16-checkpoint layers at small batch: pbl_linear_f16 dispatch (GEMV / matrix-core kernel) vs GEMV passes
vs the dense workspace + library GEMM path."""
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
    for M in (4, 8, 16, 32):
        x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
        res = {}
        def run_dispatch():                      # pbl_linear_f16: GEMV for M <= 4, matrix-core kernel above
            for l in layers: l(x)
        def run_gemv_passes():                   # the GEMV alone, 4 tokens per weight pass
            for l in layers:
                for m0 in range(0, M, 4): l(x[m0:m0 + 4])
        res["dispatch"] = round(timeit(run_dispatch, 3) / ncopy, 1)
        res["gemv_passes"] = round(timeit(run_gemv_passes, 3) / ncopy, 1)
        Q.MFMA_MAX, Q.GEMM_THRESHOLD = 0, 1       # force unpack + library GEMM
        res["dense_lib_gemm"] = round(timeit(run_dispatch, 3) / ncopy, 1)
        Q.MFMA_MAX, Q.GEMM_THRESHOLD = 32, 12
        print(json.dumps(dict(shape=shp, low_frac=lf, M=M, us_per_call=res, gflops_dispatch=round(2.0 * N * K * M / res["dispatch"] / 1e3))), flush=True)
