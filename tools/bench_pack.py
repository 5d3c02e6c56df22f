#!/usr/bin/env python3
"""Packing one 4096x4096 layer (low_frac 0.9): host packer (pbl_pack_dense_f32, all host threads, incl. the device -> host copy
of the dense weight and the host -> device copy of the blob that the QAT-eval / to_pb() paths paid in round 1) vs the device
packer (pbl_pack_dev_count + pbl_pack_dev_write)."""
import json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth
from pb_llm_amd.packing import pack_dense, pack_dense_dev

out = {"host_cpus": os.cpu_count()}
for shp, lf in (("4096x4096", 0.9), ("11008x4096", 0.95)):
    N, K = map(int, shp.split("x"))
    W = synth.llm_weight(N, K, seed=1)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    hi, lo = (r["scale"][0] + r["mean"][0]).reshape(-1), (-r["scale"][0] + r["mean"][0]).reshape(-1)
    sal = (~mask).astype(np.uint8)
    dW, dhi, dlo = torch.from_numpy(r["W_fq"]).cuda(), torch.from_numpy(hi).cuda(), torch.from_numpy(lo).cuda()
    dss, dsz, dsal = torch.from_numpy(np.asarray(r["hscale"], np.float32).reshape(-1)).cuda(), torch.from_numpy(np.asarray(r["hzero"], np.float32).reshape(-1)).cuda(), torch.from_numpy(sal).cuda()
    best_h = best_d = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.time()
        p = pack_dense(dW.cpu().numpy(), hi, lo, r["hscale"], r["hzero"], dsal.cpu().numpy()).to("cuda")
        torch.cuda.synchronize(); best_h = min(best_h, time.time() - t)
        t = time.time()
        q = pack_dense_dev(dW, dhi, dlo, dss, dsz, dsal)
        torch.cuda.synchronize(); best_d = min(best_d, time.time() - t)
    assert np.array_equal(p.blob.cpu().numpy(), q.blob.cpu().numpy())
    out[shp] = {"host_route_ms": round(best_h * 1e3, 2), "device_packer_ms": round(best_d * 1e3, 2), "blob_MB": round(q.nbytes / 1e6, 2)}
print(json.dumps(out))
