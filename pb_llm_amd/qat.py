"""QAT training step of the partially-binarized layer on the GPU.

The reference rebuilds the dense simulated weight with ~8 torch elementwise / boolean-index passes
on every forward (quant/outlier_quantizer.py:83-99) and lets autograd keep it alive for the
backward.  Here the weight-side work is three fused streaming HIP kernels behind the C ABI
(pbl_qat_scale / pbl_qat_wsim / pbl_qat_wgrad, csrc/pbl_qat.hip), w_sim is written directly in
the GEMM dtype (fp32 master weights -> bf16 under autocast, qat/run_qat.py:120) and REBUILT in the
backward instead of being saved, and the step's three GEMMs are library GEMMs.  No host
synchronisation: binary_scale stays a device scalar.
"""
from __future__ import annotations


import torch
import torch.nn.functional as F

from . import _lib

_DT = {torch.float32: _lib.PBL_DTYPE_F32, torch.float16: _lib.PBL_DTYPE_F16, torch.bfloat16: _lib.PBL_DTYPE_BF16}


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PblError("the QAT step needs GPU tensors: the HIP kernels are the only compute path")


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _mask_u8(mask: torch.Tensor) -> torch.Tensor:
    if mask.dtype not in (torch.bool, torch.uint8):
        raise TypeError("outlier mask must be a bool / uint8 tensor")
    return mask.contiguous()


def binary_scale(W: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """mean |W[~mask]| as a float32 DEVICE scalar [1] (quant/outlier_quantizer.py:90-93)."""
    _need_gpu(W, mask)
    W, mask = W.detach().contiguous(), _mask_u8(mask)
    # scratch for the two-stage reduction: a few KiB from the caching allocator per call (stream-ordered, so concurrent
    # streams never share it -- the C ABI's "no global state" holds above it as well)
    ws = torch.empty(_lib.lib().pbl_qat_workspace_bytes(), dtype=torch.uint8, device=W.device)
    out = torch.empty(1, dtype=torch.float32, device=W.device)
    _lib.check(_lib.lib().pbl_qat_scale(W.data_ptr(), _DT[W.dtype], mask.data_ptr(), W.numel(), ws.data_ptr(),
                                        out.data_ptr(), _stream(W)), "qat_scale")
    return out


def build_wsim(W: torch.Tensor, mask: torch.Tensor, scale: torch.Tensor, outlier_scale: float, out_dtype=None) -> torch.Tensor:
    """w_sim = where(mask, W*outlier_scale, sign(W)*scale) in out_dtype (quant/outlier_quantizer.py:94-98)."""
    _need_gpu(W, mask, scale)
    W, mask = W.detach().contiguous(), _mask_u8(mask)
    out = torch.empty(W.shape, dtype=out_dtype or W.dtype, device=W.device)
    _lib.check(_lib.lib().pbl_qat_wsim(W.data_ptr(), _DT[W.dtype], mask.data_ptr(), scale.data_ptr(), float(outlier_scale),
                                       out.data_ptr(), _DT[out.dtype], W.numel(), _stream(W)), "qat_wsim")
    return out


def wgrad_(g: torch.Tensor, mask: torch.Tensor, scale: torch.Tensor, outlier_scale: float, train_outlier: bool) -> torch.Tensor:
    """in place: dL/dw_sim -> dL/dW (straight-through estimator, quant/quantizer.py:18-25)."""
    _need_gpu(g, mask, scale)
    assert g.is_contiguous()
    _lib.check(_lib.lib().pbl_qat_wgrad(g.data_ptr(), _DT[g.dtype], _mask_u8(mask).data_ptr(), scale.data_ptr(),
                                        float(outlier_scale), int(bool(train_outlier)), g.numel(), _stream(g)), "qat_wgrad")
    return g


class PBQatLinearFn(torch.autograd.Function):
    """y = F.linear(x, where(mask, W*os, STE(sign(W))*s), bias) with the reference's gradients."""

    @staticmethod
    def forward(ctx, x, W, bias, mask, scale, outlier_scale, train_outlier):
        amp = torch.is_autocast_enabled("cuda")
        gdt = torch.get_autocast_dtype("cuda") if amp else W.dtype
        if not amp and x.dtype != W.dtype:
            raise RuntimeError(f"expected x and weight to have the same dtype, got {x.dtype} and {W.dtype}")  # F.linear's rule
        w_sim = build_wsim(W, mask, scale, outlier_scale, gdt)
        xg = x.to(gdt)
        with torch.autocast("cuda", enabled=False):
            y = F.linear(xg, w_sim, None if bias is None else bias.to(gdt))
        ctx.save_for_backward(xg, W, mask, scale)         # w_sim is NOT kept: rebuilt in backward
        ctx.cfg = (float(outlier_scale), bool(train_outlier), bias is not None and bias.dtype, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xg, W, mask, scale = ctx.saved_tensors
        outlier_scale, train_outlier, bias_dtype, x_dtype = ctx.cfg
        N, K = W.shape
        dy2 = dy.reshape(-1, N).to(xg.dtype)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            w_sim = build_wsim(W, mask, scale, outlier_scale, xg.dtype)
            dx = (dy2 @ w_sim).reshape(xg.shape).to(x_dtype)
        if ctx.needs_input_grad[1]:
            g = dy2.t() @ xg.reshape(-1, K)
            g = g.to(W.dtype) if g.dtype != W.dtype else g
            dW = wgrad_(g.contiguous(), mask, scale, outlier_scale, train_outlier)
        if ctx.needs_input_grad[2] and bias_dtype:
            db = dy2.sum(0).to(bias_dtype)
        return dx, dW, db, None, None, None, None


def qat_linear(x, W, bias, mask, outlier_scale=1.0, train_outlier=False):
    """One training-mode forward of BinaryXnorExceptOutliersLinear (quant/outlier_quantizer.py:83-106).
    Returns (y, binary_scale as a float32 device scalar)."""
    _need_gpu(x, W, mask, bias)
    s = binary_scale(W, mask)
    return PBQatLinearFn.apply(x, W, bias, mask, s, outlier_scale, train_outlier), s


class STEBinary(torch.autograd.Function):
    """sign() forward, identity backward (quant/quantizer.py:18-25)."""

    @staticmethod
    def forward(ctx, w):
        return w.sign()

    @staticmethod
    def backward(ctx, g):
        return g
