"""Salient selection and the 8-bit row quantizer on the GPU (the producer in front of the PB layer).

gen_outlier_mask of the reference (quant/outlier_quantizer.py:54-81) = two whole-tensor torch.kthvalue
+ mask + binary_scale + weight_quant_8bit; here each piece is a HIP kernel behind the C ABI
(csrc/pbl_prep.hip, csrc/pbl_qat.hip), bit-identical to the reference's CPU arithmetic, with no host
round trip between the pieces.
"""
from __future__ import annotations

import torch

from . import _lib
from .qat import _DT, _need_gpu, _stream, binary_scale

QUANT8_MAX_K = 16384     # pbl_quant8_rows stages a whole row in LDS


def _ws(device) -> torch.Tensor:
    """radix-select scratch, per call from the caching allocator (stream-ordered: no state shared between streams)"""
    return torch.empty(_lib.lib().pbl_prep_workspace_bytes(), dtype=torch.uint8, device=device)


def kth_pair(W: torch.Tensor, k_lo: int, k_hi: int) -> torch.Tensor:
    """[k_lo-th smallest, k_hi-th smallest] of W.flatten() (1-based, torch.kthvalue) as float32 device tensor [2]."""
    _need_gpu(W)
    W = W.detach().contiguous()
    if not (1 <= k_lo <= W.numel() and 1 <= k_hi <= W.numel()):
        raise IndexError("kthvalue(): selected number k out of range for dimension 0")     # what torch raises
    out = torch.empty(2, dtype=torch.float32, device=W.device)
    _lib.check(_lib.lib().pbl_kth_pair(W.data_ptr(), _DT[W.dtype], W.numel(), k_lo, k_hi, _ws(W.device).data_ptr(),
                                       out.data_ptr(), _stream(W)), "kth_pair")
    return out


def outlier_mask(W: torch.Tensor, thr2: torch.Tensor) -> torch.Tensor:
    """(W < thr2[0]) | (W > thr2[1]) as a bool tensor (quant/outlier_quantizer.py:69)."""
    _need_gpu(W, thr2)
    W = W.detach().contiguous()
    mask = torch.empty(W.shape, dtype=torch.bool, device=W.device)
    _lib.check(_lib.lib().pbl_outlier_mask(W.data_ptr(), _DT[W.dtype], W.numel(), thr2.data_ptr(), mask.data_ptr(), _stream(W)),
               "outlier_mask")
    return mask


def quant8_rows_(W: torch.Tensor):
    """weight_quant_8bit(W) in place (quant/outlier_quantizer.py:10-29); returns (code_scale [N], code_zp [N]) float32."""
    _need_gpu(W)
    assert W.dim() == 2 and W.is_contiguous()
    N, K = W.shape
    sc = torch.empty(N, dtype=torch.float32, device=W.device)
    zp = torch.empty(N, dtype=torch.float32, device=W.device)
    _lib.check(_lib.lib().pbl_quant8_rows(W.data_ptr(), _DT[W.dtype], N, K, sc.data_ptr(), zp.data_ptr(), _stream(W)), "quant8_rows")
    return sc, zp


def gen_outlier_mask_magnitude_(W: torch.Tensor, outlier_fraction: float):
    """gen_outlier_mask on a GPU weight: returns (mask bool [N,K], binary_scale [1,1] in W's dtype, code_scale, code_zp)
    and replaces W by its 8-bit fake quantisation IN PLACE, in the reference's order: thresholds and binary_scale from the
    original weights (:57-74), then the row quantizer (:75)."""
    n = W.numel()
    thr = kth_pair(W, int(n * outlier_fraction / 2), int(n * (1 - outlier_fraction / 2)))
    mask = outlier_mask(W, thr)
    s = binary_scale(W, mask).to(W.dtype).view(1, 1)
    if W.shape[1] <= QUANT8_MAX_K:
        sc, zp = quant8_rows_(W)
    else:
        # rows longer than the kernel stages in LDS (K = 17920 / 28672 down_proj layers): the same arithmetic in torch on
        # the device.  torch's GPU division is not correctly rounded, so a code can differ from the host reference by 1 in
        # rare ties; the kernel path is the bit-exact one.
        rng = (W.max(-1, keepdim=True)[0] - W.min(-1, keepdim=True)[0]).type(torch.float32)
        zp2 = torch.round(W.min(-1, keepdim=True)[0])
        q = torch.round((W - zp2) / rng * 255).to(torch.int64).bitwise_and(255).to(torch.uint8)   # the wrapping uint8 cast
        W.copy_((q * (rng / 255) + zp2).to(W.dtype))
        sc, zp = (rng / 255).reshape(-1), zp2.float().reshape(-1)
    return mask, s, sc, zp
