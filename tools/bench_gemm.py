#!/usr/bin/env python3
"""GEMM regime (prefill) on the llama-7b linear shapes: pbl_gemm_f16 (weight tiles rebuilt in LDS from the packed records)
vs pbl_unpack_dev + library GEMM vs the library GEMM on a resident dense fp16 weight (what the reference runs)."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q

M = int(os.environ.get("PBL_BENCH_M", 2048))
METRIC = os.environ.get("PBL_BENCH_METRIC", "magnitude")      # "hessian": BASELINE configs[2]'s column-concentrated salients (tests/cfg_shapes.py)
SHAPES = tuple((s_, float(f)) for s_, f in (t.split(":") for t in os.environ.get(
    "PBL_BENCH_SHAPES", "4096x4096:0.95,11008x4096:0.95,4096x11008:0.95").split(",")))
ONLY = os.environ.get("PBL_BENCH_ONLY", "")


PREHEAT_S = float(os.environ.get("PBL_BENCH_PREHEAT_S", 1.0))     # the same launch for this long before timing: a cold MI355X idles its clocks


def timeit(fn, n=50):
    import time
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < PREHEAT_S:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for shp, lf in SHAPES:
    N, K = map(int, shp.split("x"))
    if METRIC == "hessian":
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from cfg_shapes import hessian_layer
        W, mask, r = hessian_layer(N, K, lf, seed=300 + N % 97)
    else:
        W = synth.llm_weight(N, K, seed=N % 97)
        mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to("cuda:0")
    Wd = W16.cuda()
    x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
    flops = 2.0 * M * N * K
    res = {}
    if ONLY in ("", "fused"):
        res["fused_us"] = round(timeit(lambda: Q.fused_gemm_forward(layer.packed, None, x)), 1)
        lst = Q.gemm_list(layer.packed)              # the list kept per layer (pbl_gemm_prepare once, pbl_gemm_f16_prepared per call)
        if lst is not None:
            res["fused_prepared_us"] = round(timeit(lambda: Q.fused_gemm_forward(layer.packed, None, x, prepared=lst)), 1)
        img = Q.gemm_image(layer.packed)             # round 4: the kernel over the layer's GEMM image (built once, kept)
        if img is not None:
            res["fused_image_us"] = round(timeit(lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img)), 1)
            res["image_MB"] = round(img.data.numel() / 1e6, 2); res["image_max_slot_KiB"] = img.max_slot_kib
            # round 5: the same kernel with a thin last round cut off and split along K (pbl_gemm_f16_image_ws: what the module runs)
            import ctypes as C
            from pb_llm_amd import _lib
            plan = (C.c_uint64 * 6)()
            _lib.lib().pbl_gemm_image_plan(C.byref(layer.packed.layer_struct(None)), M, plan)
            res["plan"] = list(plan)
            if plan[0]:
                res["fused_image_split_us"] = round(timeit(lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img, split_k=True)), 1)
    if ONLY in ("", "library"):
        Q.GEMM_BACKEND = "library"
        res["unpack_plus_library_us"] = round(timeit(lambda: layer(x)), 1)
        Q.GEMM_BACKEND = "fused"
        res["dense_library_us"] = round(timeit(lambda: torch.nn.functional.linear(x, Wd)), 1)
    out = dict(shape=shp, low_frac=lf, metric=METRIC, M=M, preheat_s=PREHEAT_S, us=res, tflops={k.replace("_us", ""): round(flops / v / 1e6, 1) for k, v in res.items() if k.endswith("_us") and v})
    print(json.dumps(out), flush=True)
