#!/usr/bin/env python3
"""Sweep the launch plans of pbl_gemm_f16_image_ws (forced through pbl_debug_force_gemm_plan) on one shape: which cut / split count
is fastest, against the plan the cost model picks.  PBL_BENCH_SHAPE=5120x5120 PBL_BENCH_M=2048 PBL_PLANS="2,32,4;1,6,3;..." """
import ctypes as C, json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q, _lib

N, K = map(int, os.environ.get("PBL_BENCH_SHAPE", "5120x5120").split("x"))
M = int(os.environ.get("PBL_BENCH_M", 2048))
PLANS = [tuple(int(v) for v in p.split(",")) for p in os.environ.get("PBL_PLANS", "0,0,0;2,32,2;2,32,3;2,32,4;2,32,5;1,6,3;1,6,4;1,7,4;1,7,2").split(";")]
L = _lib.lib()
force = L.pbl_debug_force_gemm_plan; force.restype, force.argtypes = None, [C.c_int, C.c_int, C.c_int]


def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 0.6:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


W = synth.llm_weight(N, K, seed=N % 97)
mask = O.ptq_low_mask(W, 0.95, "magnitude", None, -1)
r = O.ptq_rtn(W, mask, 8, -1)
layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to("cuda:0")
x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
img = Q.gemm_image(layer.packed)
plan = (C.c_uint64 * 6)()
L.pbl_gemm_image_plan(C.byref(layer.packed.layer_struct(None)), M, plan)
out = {"shape": f"{N}x{K}", "M": M, "model_plan": list(plan), "model_plan_us": round(timeit(lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img, split_k=True)), 1)}
try:
    for p in PLANS:
        force(*p)
        L.pbl_gemm_image_plan(C.byref(layer.packed.layer_struct(None)), M, plan)
        out[f"plan_{p[0]}_{p[1]}_{p[2]}"] = (round(timeit(lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img, split_k=True)), 1), list(plan)[:4])
finally:
    force(-1, 0, 0)
print(json.dumps(out), flush=True)
