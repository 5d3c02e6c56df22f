#!/usr/bin/env python3
"""What column-concentrated (hessian) salients cost the GEMV, MEASURED ("or DESIGN.md section 9 states it closed with
the decode-step cost measured in the GEMV (not estimated)").  Two 4096 x 4096 layers at low_frac 0.95 from the product's GPU producer
-- magnitude salients (0.13 % of the column steps exceed 127) and hessian salients with 1 % hot calibration channels (3.9 %: chunks
close early, 1.5 x the chunks per row) -- as streams of L device copies each (beyond the Infinity Cache), one grouped launch per step:
us per layer, packed bytes per layer, bytes per salient entry, chunks per row, achieved GB/s over the PACKED bytes."""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from pb_llm_amd import quant as Q
from pb_llm_amd.ptq import LowHighGPTQ
from pb_llm_amd.runtime import GroupedGemv

dev = "cuda:0"
N = K = 4096
L = int(os.environ.get("PBL_BENCH_LAYERS", 96))
lf = float(os.environ.get("PBL_BENCH_LOW_FRAC", 0.95))
out = []
for metric in ("magnitude", "hessian"):
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=torch.float16)
    lin.weight.data = (torch.randn(N, K, device=dev, generator=gen) * 0.02).half()
    g = LowHighGPTQ(lin, metric, -1, 8, disable_gptq=True)
    X = torch.randn(1024, K, device=dev, generator=gen)
    hot = torch.randperm(K, device=dev, generator=gen)[:K // 100]
    X[:, hot] *= 20.0
    g.add_batch(X); g.fasterquant(lf)
    p = g.to_pb().packed; g.free()
    copies = [type(p)(p.blob.clone(), p.N, p.K, p.P, p.G, p.NRB, p.flags, p.max_nch, p.max_nexc, p.nnz, p.nexc) for _ in range(L)]
    grp = GroupedGemv(copies, None, 1, dev)
    for xi in grp.x:
        xi.copy_(torch.randn(1, K, device=dev, generator=gen).half())
    grp.launch(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(10): grp.launch()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): grp.launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50 / L
    import numpy as np
    hdr = p.blob[:80].cpu().numpy()
    info = p.blob[80:80 + 16 * (p.NRB + 1)].cpu().numpy().view(np.uint32).reshape(-1, 4)
    nch = int(info[:p.NRB, 1].sum() + info[:p.NRB, 2].sum())
    out.append(dict(metric=metric, low_frac=lf, layers=L, us_per_layer=round(us, 3), packed_MB=round(p.nbytes / 1e6, 3), nnz=int(p.nnz),
                    bytes_per_salient_entry=round((p.nbytes - N * K / 8) / max(1, int(p.nnz)), 3), chunks_per_row=round(nch / N, 2),
                    packed_GBps=round(p.nbytes / us / 1e3, 1), algorithmic_GBps=round(p.algorithmic_bytes(1) / us / 1e3, 1)))
    print(json.dumps(out[-1]), flush=True)
    del grp, copies
    torch.cuda.empty_cache()
m, h = out
print(json.dumps(dict(summary=True, hessian_over_magnitude_time=round(h["us_per_layer"] / m["us_per_layer"], 3),
                      hessian_over_magnitude_bytes=round(h["packed_MB"] / m["packed_MB"], 3),
                      note="an ideal long-gap escape brings the hessian layer's bytes and chunk count to the magnitude layer's: the time it could save is "
                           "at most the first ratio - 1")), flush=True)
