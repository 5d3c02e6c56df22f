#!/usr/bin/env python3
"""The GEMM-image kernel alone on one shape (device-synthesised layer: seconds), for A/B builds with parts of the loop removed
(tools/build_variant.sh <name> -DPBL_IMG_ABLATE=<bits>: 1 no expansion, 2 no x staging, 4 no MFMA; PBL_LIB=build/libpbl_<name>.so).
Results of ablated builds are wrong by construction -- only the time is read."""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from pb_llm_amd import quant as Q
from pb_llm_amd.ptq import LowHighGPTQ

M = int(os.environ.get("PBL_BENCH_M", 2048))
N, K = map(int, os.environ.get("PBL_BENCH_SHAPE", "4096x4096").split("x"))
lf = float(os.environ.get("PBL_BENCH_LOW_FRAC", 0.95))
dev = "cuda:0"
gen = torch.Generator(device=dev); gen.manual_seed(5)
lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=torch.float16)
lin.weight.data = (torch.randn(N, K, device=dev, generator=gen) * 0.02).half()
g = LowHighGPTQ(lin, "magnitude", -1, 8, disable_gptq=True)
g.add_batch(torch.randn(256, K, device=dev, generator=gen)); g.fasterquant(lf)
layer = g.to_pb(); g.free()
img = Q.gemm_image(layer.packed)
x = (torch.randn(M, K, device=dev, generator=gen)).half()
fn = lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img)
fn(); torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(20): fn()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
def timed(f):
    f(); torch.cuda.synchronize()
    t1 = time.time()
    while time.time() - t1 < 0.7:
        for _ in range(20): f()
        torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(50): f()
    a1.record(); torch.cuda.synchronize()
    return a0.elapsed_time(a1) * 1e3 / 50


res = {"lib": os.environ.get("PBL_LIB", "default"), "shape": f"{N}x{K}", "M": M, "low_frac": lf, "image_kernel_us": round(us, 1),
       "tflops": round(2.0 * M * N * K / us / 1e6, 1)}
if hasattr(Q, "x_fragments") and os.environ.get("PBL_BENCH_XF", "1") == "1":      # round 6: x as a fragment-major copy
    xf = Q.x_fragments(x)
    ok = bool(torch.equal(Q.fused_gemm_forward(layer.packed, None, x, image=img, x_frag=xf), Q.fused_gemm_forward(layer.packed, None, x, image=img)))
    res["xf_kernel_us"] = round(timed(lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img, x_frag=xf)), 1)
    res["xf_copy_us"] = round(timed(lambda: Q.x_fragments(x)), 1)
    res["xf_copy_plus_kernel_us"] = round(timed(lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img, x_frag=True)), 1)
    res["xf_bit_identical"] = ok
    Wd = layer.weight.half()
    res["dense_library_us"] = round(timed(lambda: torch.nn.functional.linear(x, Wd)), 1)
print(json.dumps(res), flush=True)
