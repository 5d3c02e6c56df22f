#!/usr/bin/env python3
"""<= 32 rows of x: the small-batch kernel over the GEMM image (pbl_gemm_small_image_ws) vs the round-2/3 kernel over the packed
records (pbl_gemm_mfma_f16_ws), per layer shape; PBL_SB_WAVES="1024,2048,..." sweeps the K-split target."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q, _lib

SHAPES = tuple((s_, float(f)) for s_, f in (t.split(":") for t in os.environ.get(
    "PBL_BENCH_SHAPES", "13824x5120:0.8,5120x13824:0.8,4096x4096:0.9,11008x4096:0.9").split(",")))
MS = [int(m) for m in os.environ.get("PBL_BENCH_MS", "32,16,8").split(",")]
WAVES = [int(w) for w in os.environ.get("PBL_SB_WAVES", "2048").split(",")]
L = _lib.lib()
setw = L.pbl_debug_set_small_image_waves; setw.restype = None; setw.argtypes = [C.c_int]
setp = L.pbl_debug_set_small_image_plan; setp.restype = None; setp.argtypes = [C.c_int, C.c_int]
# round 6: PBL_SB_PLANS="default,0:0,1:1,2:3,..." = geometry:ks pairs forced through pbl_debug_set_small_image_plan (default: the rule)
PLANS = [t for t in os.environ.get("PBL_SB_PLANS", "default").split(",") if t]


def timeit(fn, n=100):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 0.5:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for shp, lf in SHAPES:
    N, K = map(int, shp.split("x"))
    if os.environ.get("PBL_BENCH_SYNTH", "device") == "device":      # the product's GPU producer (seconds instead of half a minute per layer)
        from pb_llm_amd.ptq import LowHighGPTQ
        gen = torch.Generator(device="cuda:0"); gen.manual_seed(N % 97)
        lin = torch.nn.Linear(K, N, bias=False, device="cuda:0", dtype=torch.float16)
        lin.weight.data = (torch.randn(N, K, device="cuda:0", generator=gen) * 0.02).half()
        g = LowHighGPTQ(lin, "magnitude", -1, 8, disable_gptq=True)
        g.add_batch(torch.randn(256, K, device="cuda:0", generator=gen)); g.fasterquant(lf)
        layer = g.to_pb(); W16 = lin.weight.data.clone(); g.free(); del g
    else:
        W = synth.llm_weight(N, K, seed=N % 97)
        mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        W16 = torch.from_numpy(r["W_fq"]).half()
        layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to("cuda:0")
    img = Q.gemm_image(layer.packed)
    Wd = W16.cuda()
    blob_mb = layer.packed.blob.numel() / 1e6
    for M in MS:
        x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
        res = dict(shape=shp, low_frac=lf, M=M, blob_MB=round(blob_mb, 2), image_MB=round(img.data.numel() / 1e6, 2) if img else None)
        Q.GEMM_KEEP_IMAGE = "0"
        res["records_us"] = round(timeit(lambda: Q.mfma_forward(layer.packed, None, x)), 2) if M <= 32 else None
        Q.SMALL_BATCH_IMAGE = "0"
        res["routed_us"] = round(timeit(lambda: Q._pb_linear_forward(layer.packed, None, x, False, None)), 2)      # pbl_linear_f16_ws: GEMV passes or the records kernel
        res["dense_us"] = round(timeit(lambda: torch.nn.functional.linear(x, Wd)), 2)
        if img is not None:
            yref = torch.nn.functional.linear(x.float(), Wd.float())
            for w in WAVES:
                setw(w)
                y = Q.small_image_forward(layer.packed, None, x, img)
                err = float((y.float() - yref).abs().max() / yref.abs().max())
                y2 = Q.small_image_forward(layer.packed, None, x, img)
                res[f"image_us_w{w}"] = round(timeit(lambda: Q.small_image_forward(layer.packed, None, x, img)), 2)
                res[f"rel_err_w{w}"] = float(f"{err:.2e}"); res[f"repeat_w{w}"] = bool(torch.equal(y, y2))
            setw(0)
            for pl in PLANS:
                if pl == "default":
                    setp(-1, 0)
                else:
                    setp(*[int(v) for v in pl.split(":")])
                y = Q.small_image_forward(layer.packed, None, x, img)
                err = float((y.float() - yref).abs().max() / yref.abs().max())
                res[f"plan_{pl}_us"] = round(timeit(lambda: Q.small_image_forward(layer.packed, None, x, img)), 2)
                res[f"plan_{pl}_err"] = float(f"{err:.2e}")
            setp(-1, 0)
        print(json.dumps(res), flush=True)
