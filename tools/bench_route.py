#!/usr/bin/env python3
"""Small-batch routing: for M tokens, ceil(M/4) passes of the GEMV (weights streamed once per 4 tokens) against one pass of the
matrix-core kernel (once per 32 tokens, but with the per-record expansion cost).  Times both on cached packed layers."""
import ctypes as C, json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q, _lib
from pb_llm_amd.packing import PackedWeight

SHAPES = os.environ.get("PBL_BENCH_SHAPES", "4096x4096:0.9,13824x5120:0.8,11008x4096:0.95,4096x11008:0.9").split(",")
MS = [int(m) for m in os.environ.get("PBL_BENCH_M", "5,8,12,16,24,32").split(",")]
out = {}
L = _lib.lib()
for spec in SHAPES:
    shp, lf = spec.split(":"); N, K = map(int, shp.split("x")); lf = float(lf)
    W = synth.llm_weight(N, K, seed=N % 97)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    pk = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).packed
    ncopy = max(2, int(0.6e9 / (N * K * 0.3)))
    layers = [pk.to("cuda:0") for _ in range(ncopy)]
    structs = [l.layer_struct(None) for l in layers]
    st = torch.cuda.current_stream().cuda_stream
    for M in MS:
        x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
        y = torch.empty(M, N, dtype=torch.float16, device="cuda")
        def gemv():
            for s_ in structs:
                for m0 in range(0, M, 4):
                    mb = min(4, M - m0)
                    L.pbl_linear_f16(C.byref(s_), x.data_ptr() + m0 * K * 2, y.data_ptr() + m0 * N * 2, mb, 0, st)
        def mfma():
            for l in layers: Q.mfma_forward(l, None, x)
        res = {}
        for name, fn in (("gemv", gemv), ("mfma", mfma)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn()
            e1.record(); torch.cuda.synchronize()
            res[name] = round(e0.elapsed_time(e1) * 1e3 / 3 / ncopy, 1)
        out[f"{spec}/M{M}"] = res
    print(json.dumps({k: v for k, v in out.items() if k.startswith(spec)}), flush=True)
