"""GPTQ-PB post-training quantisation of one linear layer on the GPU (the producer of the PB layer).

Counterpart of gptq_pb/gptq.py (LowHighGPT), gptq_pb/low_quant.py ("xnor") and gptq_pb/high_quant.py
(per-channel asymmetric min/max) as configured by gptq_pb/run.py:127-168.  The reference walks the K columns
in a Python loop (~15 torch launches per column); here a 128-column block is ONE fused HIP launch
(pbl_gptq_block, csrc/pbl_prep.hip) followed by one library GEMM for the trailing update, the salient
threshold is an exact radix select (pbl_kth_pair) instead of a full sort, and the Hessian / Cholesky chain are
library calls.  The result can be handed straight to the packer (`to_pb`), so no dense fake-quant checkpoint
has to exist in between.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib
from .prep import kth_pair
from .qat import _need_gpu, _stream

BLOCK = 128


def gptq_blocks_(W: torch.Tensor, U: torch.Tensor, low_mask: torch.Tensor, hscale: torch.Tensor, hzero: torch.Tensor,
                 maxq: float, mean: torch.Tensor, scale: torch.Tensor, groupsize: int, feedback: bool = True) -> torch.Tensor:
    """The blocked column loop of fasterquant (gptq.py:116-168) IN PLACE on W [N,K] fp32: one fused launch per
    128-column block + one library GEMM for the trailing update.  mean/scale: [G,N,1].  Returns the per-row loss."""
    _need_gpu(W, U, low_mask)
    assert W.dtype == torch.float32 and W.is_contiguous() and U.is_contiguous() and low_mask.is_contiguous()
    N, K = W.shape
    L = _lib.lib()
    losses = torch.zeros(N, device=W.device)
    err = torch.empty(N, BLOCK, device=W.device)
    hscale, hzero = hscale.reshape(-1).contiguous(), hzero.reshape(-1).contiguous()
    for c0 in range(0, K, BLOCK):
        c1 = min(c0 + BLOCK, K)
        g = c0 // groupsize
        mg, sg = mean[g].reshape(-1).contiguous(), scale[g].reshape(-1).contiguous()
        _lib.check(L.pbl_gptq_block(W.data_ptr(), N, K, c0, c1 - c0, U.data_ptr(), low_mask.data_ptr(), hscale.data_ptr(),
                                    hzero.data_ptr(), float(maxq), mg.data_ptr(), sg.data_ptr(), err.data_ptr(),
                                    losses.data_ptr(), int(feedback), _stream(W)), "gptq_block")
        if feedback and c1 < K:
            W[:, c1:] -= err[:, :c1 - c0].matmul(U[c0:c1, c1:])                        # gptq.py:166
    return losses


CHOL_DTYPE = torch.float64      # precision of the H -> chol -> inverse -> chol(upper) chain in fasterquant (see there)


class LowHighGPTQ:
    """LowHighGPT(layer, low_quantizer("xnor", groupsize), high_quantizer(bits, perchannel, asym), salient_metric,
    disable_gptq) -- gptq_pb/gptq.py:15-33, with the two quantizers folded in."""

    def __init__(self, layer: nn.Linear, salient_metric: str = "magnitude", groupsize: int = -1, high_bit: int = 8,
                 disable_gptq: bool = False):
        if not isinstance(layer, nn.Linear):
            raise NotImplementedError("only nn.Linear layers (the LLaMA / OPT linears of gptq_pb/run.py)")
        if salient_metric not in ("magnitude", "hessian"):
            raise NotImplementedError(salient_metric)                      # gptq.py:100-101
        self.layer = layer
        self.dev = layer.weight.device
        _need_gpu(layer.weight)
        self.rows, self.columns = layer.weight.shape
        self.H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0
        self.salient_metric = salient_metric
        self.groupsize = self.columns if groupsize == -1 else groupsize
        self.n_groups = math.ceil(self.columns / self.groupsize)
        if self.groupsize % BLOCK and self.n_groups > 1:
            raise ValueError("groupsize must be a multiple of the 128-column block")   # gptq.py:102 asserts the same
        self.maxq = float(2 ** high_bit - 1)
        self.disable_gptq = disable_gptq

    # gptq.py:35-51
    def add_batch(self, inp: torch.Tensor, out=None):
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        tmp = inp.shape[0]
        inp = inp.reshape(-1, inp.shape[-1]).t()
        self.H *= self.nsamples / (self.nsamples + tmp)
        self.nsamples += tmp
        inp = math.sqrt(2 / self.nsamples) * inp.float()
        self.H += inp.matmul(inp.t())

    def _high_calibrate(self, W):
        """HighQuantizer.calibrate(weight=True), perchannel, asymmetric, no mse search (high_quant.py:29-67,95-102);
        a HIP kernel because the integer codes depend on correctly rounded divisions."""
        scale = torch.empty(W.shape[0], device=W.device)
        zero = torch.empty(W.shape[0], device=W.device)
        _lib.check(_lib.lib().pbl_high_calibrate(W.data_ptr(), W.shape[0], W.shape[1], self.maxq, scale.data_ptr(),
                                                 zero.data_ptr(), _stream(W)), "high_calibrate")
        return scale, zero

    def fasterquant(self, low_frac: float, blocksize: int = BLOCK, percdamp: float = 0.01) -> dict:
        """gptq.py:54-187.  Writes the fake-quant weight back into the layer like the reference and keeps
        mask / quantizer state on the object (mask, mean, scale, hscale, hzero) for `to_pb`."""
        if blocksize != BLOCK:
            raise NotImplementedError("the fused block kernel is built for the reference's default blocksize 128")
        N, K = self.rows, self.columns
        W = self.layer.weight.data.clone().float().contiguous()
        hscale, hzero = self._high_calibrate(W)                                        # :62-63
        H = self.H
        del self.H
        dead = torch.diag(H) == 0                                                      # :67-70
        H[dead, dead] = 1
        W[:, dead] = 0
        damp = percdamp * torch.mean(torch.diag(H))                                    # :74-81
        idx = torch.arange(K, device=self.dev)
        H[idx, idx] += damp
        # The reference runs this chain in fp32 on the CPU (LAPACK); an fp32 chain on the GPU (rocSOLVER, other blocking and
        # summation order) lands ~1e-4 away from it, which moves hessian-metric masks and GPTQ roundings near their
        # thresholds.  In fp64, rounded ONCE to fp32, the factor is within a few fp32 ulp of the true one -- and so is the
        # reference's (measured on golden G5: 2.8e-7 of max|U| between its stored U and the fp64 chain) -- so that is what
        # pins this step (CHOL_DTYPE = torch.float32 restores the single-precision chain).
        H = H.to(CHOL_DTYPE)
        H = torch.linalg.cholesky(H)
        H = torch.cholesky_inverse(H)
        U = torch.linalg.cholesky(H, upper=True).to(torch.float32).contiguous()
        mask = torch.zeros_like(W, dtype=torch.bool)
        mean = torch.zeros(self.n_groups, N, 1, device=self.dev)
        scale = torch.zeros(self.n_groups, N, 1, device=self.dev)
        for g in range(self.n_groups):                                                 # :84-106
            st, ed = g * self.groupsize, min((g + 1) * self.groupsize, K)
            if self.salient_metric == "magnitude":
                sal = torch.abs(W[:, st:ed])
            else:
                sal = W[:, st:ed] ** 2 / (torch.diag(U[st:ed, st:ed]).reshape((1, -1))) ** 2
            sal = sal.contiguous()
            k = int(sal.numel() * low_frac)                                            # sorted[k] == (k+1)-th smallest
            if k >= sal.numel():
                raise IndexError("index out of range: low_frac selects past the end")  # the reference's sort()[k] raises
            thresh = kth_pair(sal, k + 1, k + 1)[0]
            mask[:, st:ed] = sal <= thresh
            wm = W[:, st:ed] * mask[:, st:ed]                                          # LowQuantizer.calibrate "xnor"
            mean[g] = wm.mean(-1).view(-1, 1)                                          # low_quant.py:25-32
            scale[g] = (wm - mean[g]).abs().mean(-1, keepdim=True)
        losses = gptq_blocks_(W, U, mask.contiguous(), hscale, hzero, self.maxq, mean, scale, self.groupsize,
                              feedback=not self.disable_gptq)
        self.layer.weight.data = W.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)
        self.mask, self.mean, self.scale = mask, mean, scale
        self.hscale, self.hzero = hscale.view(-1, 1), hzero.view(-1, 1)
        self.hinv_diag = torch.diag(U).clone()
        self.losses = losses
        return {"error": torch.sum(losses).item() if not self.disable_gptq else 0.0}

    def to_pb(self):
        """Pack the quantised layer (after fasterquant) into a PBLinear without going through a dense checkpoint."""
        from .quant import PBLinear
        from .packing import pack_dense_dev
        W = self.layer.weight.data
        N = W.shape[0]
        # levels from the low quantizer's state, rounded like the written-back weight (gptq.py:182 `.to(dtype)`); the matrix,
        # the mask and the quantizer state are all on the GPU, so the blob is built there (csrc/pbl_pack.hip)
        hi = (self.scale + self.mean).reshape(self.n_groups, N).t().to(W.dtype).float().contiguous()
        lo = (-self.scale + self.mean).reshape(self.n_groups, N).t().to(W.dtype).float().contiguous()
        packed = pack_dense_dev(W.float(), hi, lo, self.hscale.reshape(-1), self.hzero.reshape(-1), ~self.mask,
                                sal_f16=W.dtype == torch.float16)
        bias = None if self.layer.bias is None else self.layer.bias.data
        return PBLinear(packed, bias, W.dtype)

    def free(self):
        self.H = None
