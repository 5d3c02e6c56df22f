"""Host-side container and packer front-ends for the PBL1 format (include/pbl.h).

The reference never stores packed weights: PTQ writes a dense fake-quant fp16
matrix back into nn.Linear (gptq_pb/gptq.py:180-184) and QAT re-simulates one on
every forward (quant/outlier_quantizer.py:83-99).  `PackedWeight` is what this
build stores instead: a dense 1-bit plane + uint8-coded salient list.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


def _f32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().float().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


@dataclass
class PackedWeight:
    """A PBL1 blob (uint8 tensor, host or device) plus its header fields."""
    blob: torch.Tensor
    N: int
    K: int
    P: int
    G: int
    NRB: int
    flags: int
    max_nch: int
    max_nexc: int
    nnz: int
    nexc: int

    @property
    def nbytes(self) -> int:
        return int(self.blob.numel())

    def to(self, device) -> "PackedWeight":
        return PackedWeight(self.blob.to(device), self.N, self.K, self.P, self.G, self.NRB, self.flags,
                            self.max_nch, self.max_nexc, self.nnz, self.nexc)

    def layer_struct(self, bias: torch.Tensor | None = None) -> _lib.PblLayer:
        if self.blob.data_ptr() % 16:
            raise _lib.PblError("blob is not 16-byte aligned")
        return _lib.PblLayer(self.blob.data_ptr(), bias.data_ptr() if bias is not None else None,
                             self.N, self.K, self.P, self.G, self.NRB, self.flags, self.max_nch, self.max_nexc)

    def unpack(self) -> torch.Tensor:
        """Dense simulated weight, fp32 [N, K] on the host (to_regular_linear,
        quant/outlier_quantizer.py:108-114)."""
        host = self.blob.cpu().contiguous()
        out = np.empty((self.N, self.K), np.float32)
        _lib.check(_lib.lib().pbl_unpack_dense_f32(host.data_ptr(), host.numel(), out.ctypes.data), "unpack")
        return torch.from_numpy(out)

    def algorithmic_bytes(self, M: int = 1, with_bias: bool = False) -> int:
        """B_alg of SURVEY.md 8(d): sign plane + uint8 code + 8-bit index per salient
        + per-row(-group) (alpha,mu) fp16 + per-row salient scale/zero fp32 + CSR row
        pointers + fp16 x and y."""
        b = self.N * self.K // 8 + 2 * self.nnz + 4 * self.N * self.G + 8 * self.N + 4 * (self.N + 1)
        b += 2 * M * self.K + 2 * M * self.N + (2 * self.N if with_bias else 0)
        return int(b)

    @staticmethod
    def from_blob(blob: torch.Tensor) -> "PackedWeight":
        host = blob.cpu().contiguous()
        h = _lib.PblBlobHeader.from_buffer_copy(bytes(host[:C.sizeof(_lib.PblBlobHeader)].numpy()))
        layer = _lib.PblLayer()
        _lib.check(_lib.lib().pbl_blob_describe(host.data_ptr(), host.numel(), C.byref(layer)), "describe")
        return PackedWeight(blob, h.N, h.K, h.P, h.G, h.NRB, h.flags, h.max_nch, h.max_nexc, h.nnz, h.nexc)


def pack_dense(W, hi, lo, sscale=None, szero=None, sal_mask=None) -> PackedWeight:
    """Pack a dense simulated weight.  W [N,K]; hi, lo [N,G] (the two values the
    binarized weights of each row/group take); sscale, szero [N] (salient value =
    sscale*(q - szero), HighQuantizer's form, gptq_pb/high_quant.py:6-8) or None.
    Exact for any input: values on neither level nor the code grid are stored as
    fp32 exceptions."""
    W = _f32(W)
    N, K = W.shape
    hi = _f32(hi).reshape(N, -1)
    lo = _f32(lo).reshape(N, -1)
    G = hi.shape[1]
    if lo.shape != hi.shape:
        raise ValueError("hi / lo shape mismatch")
    ss = _f32(sscale).reshape(N) if sscale is not None else None
    sz = _f32(szero).reshape(N) if szero is not None else None
    sm = None
    if sal_mask is not None:
        sm = np.ascontiguousarray(
            sal_mask.detach().cpu().numpy() if isinstance(sal_mask, torch.Tensor) else sal_mask).astype(np.uint8)
    L = _lib.lib()
    ptr = lambda a: a.ctypes.data if a is not None else None  # noqa: E731
    size = C.c_size_t(0)
    _lib.check(L.pbl_pack_dense_f32(ptr(W), N, K, G, ptr(hi), ptr(lo), ptr(ss), ptr(sz), ptr(sm),
                                    None, 0, C.byref(size)), "pack(size)")
    blob = torch.empty(size.value, dtype=torch.uint8)
    _lib.check(L.pbl_pack_dense_f32(ptr(W), N, K, G, ptr(hi), ptr(lo), ptr(ss), ptr(sz), ptr(sm),
                                    blob.data_ptr(), size.value, C.byref(size)), "pack")
    return PackedWeight.from_blob(blob)


def infer_levels(W: np.ndarray, groupsize: int = -1, low_mask: np.ndarray | None = None):
    """Per (row, column group): the two most frequent values among the binarized
    positions -> (hi, lo) with hi >= lo.  This is how PB structure is re-discovered
    from a flattened dense checkpoint (qat/eval_after_qat.py:12-15 loads only dense
    weights)."""
    W = _f32(W)
    N, K = W.shape
    gs = K if groupsize == -1 else groupsize
    G = (K + gs - 1) // gs
    hi = np.zeros((N, G), np.float32)
    lo = np.zeros((N, G), np.float32)
    for g in range(G):
        blk = W[:, g * gs:(g + 1) * gs]
        mk = low_mask[:, g * gs:(g + 1) * gs] if low_mask is not None else None
        for r in range(N):
            vals = blk[r][mk[r]] if mk is not None else blk[r]
            if vals.size == 0:
                continue
            u, c = np.unique(vals, return_counts=True)
            order = np.argsort(-c, kind="stable")
            a = u[order[0]]
            b = u[order[1]] if u.size > 1 else a
            hi[r, g], lo[r, g] = max(a, b), min(a, b)
    return hi, lo


def infer_code_grid(W: np.ndarray, hi: np.ndarray, lo: np.ndarray, groupsize: int = -1):
    """Per-row affine grid (sscale, szero) of the values that are on neither level:
    sscale = smallest positive gap between distinct such values, szero chosen so
    the smallest code is 0.  Values that are off the inferred grid simply become
    exceptions in the packer, so a wrong guess costs bytes, never correctness."""
    W = _f32(W)
    N, K = W.shape
    gs = K if groupsize == -1 else groupsize
    hi_full = np.repeat(hi, gs, axis=1)[:, :K]
    lo_full = np.repeat(lo, gs, axis=1)[:, :K]
    other = (W != hi_full) & (W != lo_full)
    ss = np.ones(N, np.float32)
    sz = np.zeros(N, np.float32)
    for r in range(N):
        v = np.unique(W[r][other[r]])
        if v.size == 0:
            continue
        if v.size == 1:
            ss[r] = abs(v[0]) if v[0] != 0 else 1.0
            sz[r] = 0.0 if v[0] >= 0 else 2.0  # value = ss*(q - sz): q=1 or q=1 with sz=2 -> -ss
            if v[0] == 0:
                sz[r] = 0.0
            continue
        gaps = np.diff(v.astype(np.float64))
        s = float(gaps[gaps > 0].min())
        span = np.rint((v[-1] - v[0]) / s)
        if 1 <= span <= 255:
            s = float((v[-1] - v[0]) / span)
        # the true scale is an fp32 number: pick the neighbour of the estimate that
        # reproduces the most values bit-exactly as fl32(s*(q - z))
        best = (-1, np.float32(s), np.float32(0))
        cand = np.float32(s)
        cands = [cand]
        for _ in range(3):
            cands.append(np.nextafter(cands[-1], np.float32(np.inf)))
        cand = np.float32(s)
        for _ in range(3):
            cand = np.nextafter(cand, np.float32(-np.inf))
            cands.append(cand)
        for c in cands:
            z = np.float32(-np.rint(v[0] / c))
            q = np.rint(v / c) + z
            rec = (c * (q.astype(np.float32) - z)).astype(np.float32)
            hits = int(np.count_nonzero((rec == v) & (q >= 0) & (q <= 255)))
            if hits > best[0]:
                best = (hits, c, z)
        ss[r], sz[r] = best[1], best[2]
    return ss, sz
