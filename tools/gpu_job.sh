#!/bin/bash
# One gpurun call (rewritten per call; results under gpurun_out/<tag>/).  Usage: tools/gpu_job.sh <tag>
# r4-4: GPU tests touched so far this round + what the fp64 Cholesky chain buys against the reference's stored factor (G5)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r44}; mkdir -p $O
timeout 120 python __graft_entry__.py > $O/build.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pack.py tests/test_gpu_ptq.py tests/test_gpu_gemm.py -q -m gpu -x -k "bf16 or from_quantizers or g5 or gptq or fully_binarized" 2>&1 | tail -15 > $O/test_sel.txt
cat $O/test_sel.txt
cd tests && timeout 120 python - <<'P' > ../$O/chol.txt 2>&1
import numpy as np, torch, sys
sys.path.insert(0, '..')
from conftest import golden
from test_oracle_golden import g5_inputs, g5_name
from pb_llm_amd import ptq
import torch.nn as nn
W16, Xcal, x1, x32 = g5_inputs()
for metric, gs in (("magnitude", -1), ("hessian", -1), ("hessian", 128), ("magnitude", 128)):
    g = golden(g5_name(metric, gs, False, 0.9))
    for dt in (torch.float32, torch.float64):
        ptq.CHOL_DTYPE = dt
        layer = nn.Linear(768, 768, bias=False); layer.weight.data = torch.from_numpy(W16).clone(); layer = layer.cuda()
        q = ptq.LowHighGPTQ(layer, salient_metric=metric, groupsize=gs, high_bit=8, disable_gptq=False)
        for s in range(Xcal.shape[0]):
            q.add_batch(torch.from_numpy(Xcal[s:s + 1]).cuda(), None)
        info = q.fasterquant(0.9, blocksize=128, percdamp=0.01)
        gm = np.unpackbits(g["mask"])[:768 * 768].astype(bool).reshape(768, 768)
        Wq = layer.weight.data.cpu().numpy()
        print(metric, gs, dt, "mask mism", int(np.count_nonzero(q.mask.cpu().numpy() != gm)), "hinv_diag rel", float(np.abs(q.hinv_diag.cpu().numpy() / g["hinv_diag"] - 1).max()),
              "W equal frac", float(np.mean(Wq == g["W_fq"])), "loss rel", abs(info["error"] - float(g["loss"])) / float(g["loss"]))
P
cd ..; cat $O/chol.txt
