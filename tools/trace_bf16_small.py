#!/usr/bin/env python3
"""What a small decode batch of bf16 activations launches (round 5): `module(x_bf16)` at 16 and 48 rows on a 4096 x 4096 layer
(low_frac 0.9), 300 calls each, next to the same calls with fp16 activations.  Run under `rocprofv3 --kernel-trace --stats`
(tools/gpu_job.sh bf16trace): per call one act_bf16_prepare_kernel, one pbl_sb_img_kernel and one sb_reduce_kernel<true> (scale,
bias and bf16 cast inside the K splits' reduce); fp16: the kernel and sb_reduce_kernel<false>."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q

N = K = 4096
W = synth.llm_weight(N, K, seed=3)
mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
r = O.ptq_rtn(W, mask, 8, -1)
layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to("cuda:0")
out = {}
with torch.no_grad():
    for M in (16, 48):
        x = torch.from_numpy(synth.activations((M, K), 5, 21)).to("cuda:0")
        for name, xi in (("fp16", x), ("bf16", x.bfloat16())):
            for _ in range(20): layer(xi)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300): layer(xi)
            torch.cuda.synchronize()
            out[f"{name}_M{M}_us_per_call"] = round((time.perf_counter() - t0) / 300 * 1e6, 2)
print(out)
