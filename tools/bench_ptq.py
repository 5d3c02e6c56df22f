#!/usr/bin/env python3
"""GPTQ-PB quantisation of one LLaMA-7B-sized linear on one MI355X: the fused block pipeline (pb_llm_amd/ptq.py) vs the
reference's column loop as written (gptq_pb/gptq.py:129-168) executed with torch ops on the same GPU (timed on the first
blocks and extrapolated: it is minutes-slow by construction)."""
import json, os, sys, time
import numpy as np, torch, torch.nn as nn
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from pb_llm_amd import ptq, synth

N = K = 4096
W = torch.from_numpy(synth.llm_weight(N, K, seed=3, heavy_tail=True)).half().cuda()
X = torch.from_numpy(synth.calib_inputs(8, 512, K, seed=3)).cuda()


def fused():
    layer = nn.Linear(K, N, bias=False).cuda().half()
    layer.weight.data = W.clone()
    q = ptq.LowHighGPTQ(layer, "hessian", -1, 8, False)
    for s in range(X.shape[0]):
        q.add_batch(X[s:s + 1])
    torch.cuda.synchronize(); t0 = time.time()
    info = q.fasterquant(0.9)
    torch.cuda.synchronize(); t1 = time.time()
    return t1 - t0, info["error"], q


t_f, err_f, q = fused()
t_f, err_f, q = fused()          # second call: libraries warmed up
# as written: per-column torch ops on the GPU for the first 2 blocks of the same problem
Wf = W.float().clone(); U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(
    (lambda H: H + 0.01 * torch.mean(torch.diag(H)) * torch.eye(K, device="cuda"))(
        2.0 / X.shape[0] * sum((x.reshape(-1, K).t().float() @ x.reshape(-1, K).float()) for x in X)))), upper=True)
mask, hs, hz, mean, scale = q.mask, q.hscale, q.hzero, q.mean[0], q.scale[0]
torch.cuda.synchronize(); t0 = time.time()
nblk = 2
for c0 in range(0, 128 * nblk, 128):
    W1 = Wf[:, c0:c0 + 128].clone(); Q1 = torch.zeros_like(W1); E1 = torch.zeros_like(W1); U1 = U[c0:c0 + 128, c0:c0 + 128]
    for i in range(128):
        w = W1[:, i]; d = U1[i, i]
        qh = (hs * (torch.clamp(torch.round(w.unsqueeze(1) / hs) + hz, 0, 255) - hz)).flatten()
        ql = ((w.unsqueeze(1) - mean).sign() * scale + mean).flatten()
        qq = qh * ~mask[:, c0 + i] + ql * mask[:, c0 + i]
        Q1[:, i] = qq
        e = (w - qq) / d
        W1[:, i:] -= e.unsqueeze(1).matmul(U1[i, i:].unsqueeze(0)); E1[:, i] = e
    Wf[:, c0:c0 + 128] = Q1
    Wf[:, c0 + 128:] -= E1.matmul(U[c0:c0 + 128, c0 + 128:])
torch.cuda.synchronize(); t_ref = (time.time() - t0) / nblk * (K // 128)
print(json.dumps(dict(layer=f"{N}x{K}", metric="hessian", low_frac=0.9, fused_fasterquant_s=round(t_f, 4), loss=round(err_f, 3),
                      column_loop_as_written_torch_gpu_s=round(t_ref, 2), speedup=round(t_ref / t_f, 1))))
