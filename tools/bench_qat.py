#!/usr/bin/env python3
"""QAT step of the partially-binarized layer on one MI355X: the fused weight-side kernels (pbl_qat_*) vs the
reference's forward as written (quant/outlier_quantizer.py:83-99) executed by torch on the same GPU."""
import json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from pb_llm_amd import qat, synth


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (N, K, M) in ((4096, 4096, 2048), (11008, 4096, 2048)):
    W = torch.from_numpy(synth.llm_weight(N, K, seed=3)).cuda().requires_grad_(True)          # fp32 master weights
    mask = W.detach().abs() > W.detach().abs().flatten().kthvalue(int(0.9 * N * K))[0]
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    n = N * K

    def weight_side_fused():
        s = qat.binary_scale(W, mask)
        return qat.build_wsim(W, mask, s, 1.0, torch.bfloat16)

    def weight_side_as_written():
        s = W[~mask].abs().mean(-1).view(-1, 1).detach()
        return torch.where(mask, (W * 1.0).detach(), W.sign() * s).to(torch.bfloat16)

    def step_fused():
        W.grad = None; x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y, _ = qat.qat_linear(x, W, None, mask, 1.0, False)
        y.backward(dy)

    def step_as_written():
        W.grad = None; x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            s = W[~mask].abs().mean(-1).view(-1, 1).detach()
            w_sim = torch.where(mask, (W * 1.0).detach(), qat.STEBinary.apply(W) * s)
            y = torch.nn.functional.linear(x, w_sim, None)
        y.backward(dy)

    t_f, t_r = timeit(weight_side_fused), timeit(weight_side_as_written)
    s_f, s_r = timeit(step_fused, 10), timeit(step_as_written, 10)
    alg = n * (4 + 1) + n * (4 + 1 + 2)              # scale: W + mask; wsim: W + mask + bf16 out
    print(json.dumps(dict(shape=f"{N}x{K}", tokens=M,
                          weight_side_us=dict(fused=round(t_f, 1), as_written=round(t_r, 1)),
                          weight_side_GBps=round(alg / t_f / 1e3), hbm_frac=round(alg / t_f / 1e3 / 8000, 3),
                          step_us=dict(fused=round(s_f, 1), as_written=round(s_r, 1)),
                          step_tflops=round(6.0 * N * K * M / s_f / 1e6, 1))), flush=True)
