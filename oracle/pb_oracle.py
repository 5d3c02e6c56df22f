"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (numpy) of the PB-LLM partially-binarized linear hot path and of
the quantizer steps that define its operands.  Only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import this file; the shipped
package (pb_llm_amd/) must not, and fails loudly when its HIP library is absent.

Pinning: the reference repository has no tests or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference
itself, produced by importing /root/reference in the build container with
tools/gen_goldens.py and committed under tests/golden/ (see
tests/test_oracle_golden.py).  The arithmetic of `F.linear` lives in torch, which
the reference does not pin (README.md:35); the goldens pin it to torch 2.10 CPU.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  dtype semantics follow torch: elementwise fp32 ops are done in
np.float32 so that masks / integer codes are bit-exact; reductions (mean / sum)
are done in float64 and rounded once (torch's fp32 cascade sums differ from that
by <= a few ulp, which the golden tests allow for scales only).
"""
from __future__ import annotations

import math
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# quant/quantizer.py
# --------------------------------------------------------------------------- #
def ste_binary(w: np.ndarray) -> np.ndarray:
    """STEBinary.forward: x.sign(), with sign(0) == 0 (quant/quantizer.py:18-21)."""
    return np.sign(w).astype(w.dtype)


def binary_linear_weight(W: np.ndarray) -> np.ndarray:
    """BinaryLinear: w = sign(W) in fp32, no scale (quant/quantizer.py:76-86)."""
    return ste_binary(W.astype(F32))


def xnor_binary_linear_weight(W: np.ndarray, outlier_mask: np.ndarray | None = None) -> np.ndarray:
    """XnorBinaryLinear.quant_weight (quant/quantizer.py:181-189):
    w_c = W - rowmean(W); optional w_c *= ~outlier_mask; alpha_r = mean_j |w_c|;
    w = sign(w_c) * alpha_r.  The mean is not added back."""
    W = W.astype(F32)
    mean = W.astype(np.float64).mean(-1).astype(F32).reshape(-1, 1)
    wc = (W - mean).astype(F32)
    if outlier_mask is not None:
        wc = (wc * (~outlier_mask)).astype(F32)
    alpha = np.abs(wc).astype(np.float64).mean(-1).astype(F32).reshape(-1, 1)
    return (np.sign(wc).astype(F32) * alpha).astype(F32)


# --------------------------------------------------------------------------- #
# quant/outlier_quantizer.py
# --------------------------------------------------------------------------- #
def _float_to_uint8_wrap(r: np.ndarray) -> np.ndarray:
    """torch `.type(torch.uint8)` on a float tensor: convert through int64 and keep
    the low 8 bits, so negative values wrap mod 256 (SURVEY appendix B-1;
    c10/util/TypeCast.h).  NaN/inf go through the x86 cvttss2si sentinel
    (INT64_MIN -> 0), which numpy's astype(int64) reproduces."""
    with np.errstate(invalid="ignore"):
        return r.astype(np.int64).astype(np.uint8)


def weight_quant_8bit(w: np.ndarray, simulated: bool = True) -> np.ndarray:
    """weight_quant_8bit (quant/outlier_quantizer.py:10-29).  Per-row asymmetric
    8-bit fake quant with the reference's quirks: the zero point is ROUNDED to an
    integer (usually -0.), the uint8 cast WRAPS, and clamp(0,255) on uint8 is a
    no-op."""
    raw = w.dtype
    wmax = w.max(-1, keepdims=True)
    wmin = w.min(-1, keepdims=True)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        w_range = (wmax - wmin).astype(raw).astype(F32)          # :13-16
        zp = np.rint(wmin).astype(raw)                           # :17
        t = ((w - zp).astype(raw).astype(F32) / w_range * F32(255)).astype(F32)  # :18-20
        q = _float_to_uint8_wrap(np.rint(t))                     # :18-22
        if not simulated:
            return q                                             # :27-28
        deq = (q.astype(F32) * (w_range / F32(255)).astype(F32)).astype(F32) + zp.astype(F32)
        return deq.astype(raw)                                   # :23-26


def kthvalue(flat: np.ndarray, k: int):
    """torch.kthvalue: k-th smallest, k is 1-based."""
    if k < 1 or k > flat.size:
        raise IndexError("kthvalue: k out of range")  # torch raises (SURVEY B-8)
    return np.partition(flat, k - 1)[k - 1]


def gen_outlier_mask_magnitude(W: np.ndarray, outlier_fraction: float):
    """BinaryXnorExceptOutliersLinear.gen_outlier_mask (quant/outlier_quantizer.py:54-81).
    Returns (outlier_mask bool [N,K], binary_scale fp scalar-as-[1,1], W_hat).
    Two GLOBAL order statistics, strict comparisons; binary_scale is ONE scalar
    per tensor computed on the pre-quant weights (:72-74); then the weights are
    replaced by their 8-bit fake-quant (:75)."""
    flat = W.reshape(-1)
    n = flat.size
    lo = kthvalue(flat, int(n * outlier_fraction / 2))            # :58-62
    hi = kthvalue(flat, int(n * (1 - outlier_fraction / 2)))      # :63-66
    mask = (W < lo) | (W > hi)                                    # :69
    scale = np.abs(W[~mask]).astype(np.float64).mean().astype(W.dtype).reshape(1, 1)  # :72-74
    W_hat = weight_quant_8bit(W)                                  # :75
    return mask, scale, W_hat


def refresh_binary_scale(W_hat: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """train()-mode refresh of binary_scale from the CURRENT weights
    (quant/outlier_quantizer.py:90-93); the value persists into later eval()."""
    return np.abs(W_hat[~mask]).astype(np.float64).mean().astype(W_hat.dtype).reshape(1, 1)


def binarize_except_outliers(W_hat: np.ndarray, mask: np.ndarray, binary_scale: np.ndarray,
                             outlier_scale: float = 1.0) -> np.ndarray:
    """w_sim = where(mask, W*outlier_scale, sign(W)*binary_scale)
    (quant/outlier_quantizer.py:94-98)."""
    dt = W_hat.dtype
    scaled = (W_hat * dt.type(outlier_scale)).astype(dt)
    binary = (np.sign(W_hat).astype(dt) * binary_scale.astype(dt)).astype(dt)
    return np.where(mask, scaled, binary).astype(dt)


def calc_outlier_nbits(W_hat: np.ndarray, mask: np.ndarray) -> float:
    """calc_memory_consumption (quant/outlier_quantizer.py:116-122): uint8 codes of
    the CURRENT weights, masked, to CSR; 8 bits per column index, value and row
    pointer.  Codes that are 0 at masked positions vanish from the CSR."""
    codes = weight_quant_8bit(W_hat, simulated=False)
    w_out = codes * mask
    nnz = int(np.count_nonzero(w_out))
    n_rows = W_hat.shape[0]
    return (nnz * 8 + nnz * 8 + (n_rows + 1) * 8) / W_hat.size


# --------------------------------------------------------------------------- #
# gptq_pb/low_quant.py (xnor branch) and gptq_pb/high_quant.py (asym min-max)
# --------------------------------------------------------------------------- #
def low_xnor_calibrate(w_masked: np.ndarray):
    """LowQuantizer.calibrate, method "xnor" (gptq_pb/low_quant.py:25-32).  `w_masked`
    is W[:, group] * mask: salients are zero-filled but still counted in the
    denominators (gptq_pb/gptq.py:103-105).  Returns (mean [N,1], scale [N,1])."""
    w = w_masked.astype(F32)
    mean = w.astype(np.float64).mean(-1).astype(F32).reshape(-1, 1)
    scale = np.abs((w - mean).astype(F32)).astype(np.float64).mean(-1).astype(F32).reshape(-1, 1)
    return mean, scale


def low_xnor_quantize(w: np.ndarray, mean: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """LowQuantizer.quantize, "xnor": sign(w - mean) * scale + mean
    (gptq_pb/low_quant.py:75-82); sign(0) == 0 gives exactly `mean`."""
    w = w.astype(F32)
    return ((np.sign((w - mean).astype(F32)).astype(F32) * scale).astype(F32) + mean).astype(F32)


def high_calibrate(W: np.ndarray, bits: int = 8):
    """HighQuantizer.calibrate(weight=True, perchannel=True, sym=False, mse=False)
    (gptq_pb/high_quant.py:29-67,95-102 as configured by gptq_pb/run.py:132-137).
    Returns (scale [N,1], zero [N,1], maxq)."""
    W = W.astype(F32)
    maxq = F32(2 ** bits - 1)
    xmin = np.minimum(W.min(1), F32(0))
    xmax = np.maximum(W.max(1), F32(0))
    both0 = (xmin == 0) & (xmax == 0)
    xmin = np.where(both0, F32(-1), xmin).astype(F32)
    xmax = np.where(both0, F32(1), xmax).astype(F32)
    scale = ((xmax - xmin).astype(F32) / maxq).astype(F32)
    zero = np.rint((-xmin / scale).astype(F32)).astype(F32)
    return scale.reshape(-1, 1), zero.reshape(-1, 1), maxq


def high_quantize(w: np.ndarray, scale: np.ndarray, zero: np.ndarray, maxq) -> np.ndarray:
    """quantize() (gptq_pb/high_quant.py:6-8): scale * (clamp(round(x/scale)+zero, 0, maxq) - zero)."""
    w = w.astype(F32)
    q = np.clip((np.rint((w / scale).astype(F32)) + zero).astype(F32), F32(0), F32(maxq))
    return (scale * (q - zero).astype(F32)).astype(F32)


def high_codes(w: np.ndarray, scale: np.ndarray, zero: np.ndarray, maxq) -> np.ndarray:
    """The integer code behind high_quantize (0..maxq), as uint8."""
    w = w.astype(F32)
    q = np.clip((np.rint((w / scale).astype(F32)) + zero).astype(F32), F32(0), F32(maxq))
    return q.astype(np.uint8)


# --------------------------------------------------------------------------- #
# gptq_pb/gptq.py
# --------------------------------------------------------------------------- #
def hessian_from_inputs(X: np.ndarray) -> np.ndarray:
    """LowHighGPT.add_batch over samples X [nsamples, seq, K] (gptq_pb/gptq.py:35-51):
    running H = 2/n * sum_t x_t x_t^T with n = number of add_batch calls' batch dim
    (each call has batch 1 in gptq_pb/run.py:155-156)."""
    K = X.shape[-1]
    H = np.zeros((K, K), F32)
    ns = 0
    for s in range(X.shape[0]):
        inp = X[s].reshape(-1, K).T.astype(F32)
        H *= F32(ns / (ns + 1))
        ns += 1
        inp = (F32(math.sqrt(2 / ns)) * inp).astype(F32)
        H += inp @ inp.T
    return H


def hinv_cholesky_upper(H: np.ndarray, percdamp: float = 0.01):
    """gptq_pb/gptq.py:67-81: dead columns, damping, then
    U = chol(cholesky_inverse(chol(H)), upper).  Returns (U, dead)."""
    H = H.astype(F32).copy()
    dead = np.diag(H) == 0
    H[dead, dead] = 1
    damp = F32(percdamp) * np.mean(np.diag(H)).astype(F32)
    idx = np.arange(H.shape[0])
    H[idx, idx] += damp
    L = np.linalg.cholesky(H.astype(F32))
    Linv = np.linalg.inv(L.astype(np.float64))
    Hinv = (Linv.T @ Linv).astype(F32)
    U = np.linalg.cholesky(Hinv.astype(F32)).T.copy()
    return U.astype(F32), dead


def ptq_low_mask(W: np.ndarray, low_frac: float, metric: str = "magnitude",
                 hinv_diag: np.ndarray | None = None, groupsize: int = -1) -> np.ndarray:
    """Low (= binarized) mask, True means BINARIZED (gptq_pb/gptq.py:83-99).  One
    GLOBAL threshold per column group: sorted[int(numel*low_frac)], `<=`.
    Hessian metric: w^2 / diag(U)^2 with U the upper Cholesky factor of H^-1
    (SURVEY appendix B-11)."""
    W = W.astype(F32)
    N, K = W.shape
    gs = K if groupsize == -1 else groupsize
    mask = np.zeros((N, K), bool)
    for st in range(0, K, gs):
        ed = min(st + gs, K)
        if metric == "magnitude":
            sal = np.abs(W[:, st:ed])
        elif metric == "hessian":
            d = hinv_diag[st:ed].astype(F32).reshape(1, -1)
            sal = ((W[:, st:ed] ** 2).astype(F32) / (d ** 2).astype(F32)).astype(F32)
        else:
            raise NotImplementedError(metric)          # gptq.py:100-101
        flat = sal.reshape(-1)
        k = int(flat.size * low_frac)
        thresh = np.partition(flat, k)[k]
        mask[:, st:ed] = sal <= thresh
    return mask


def ptq_rtn(W: np.ndarray, mask: np.ndarray, high_bit: int = 8, groupsize: int = -1):
    """The `disable_gptq` (RTN) branch of LowHighGPT.fasterquant
    (gptq_pb/gptq.py:62-63,103-105,119-127): q = q_high*~mask + q_low*mask per
    128-column block.  Returns dict(W_fq, mean[G,N,1], scale[G,N,1], hscale, hzero)."""
    W = W.astype(F32)
    N, K = W.shape
    gs = K if groupsize == -1 else groupsize
    G = math.ceil(K / gs)
    hscale, hzero, maxq = high_calibrate(W, high_bit)
    mean = np.zeros((G, N, 1), F32)
    scale = np.zeros((G, N, 1), F32)
    out = np.empty_like(W)
    for g in range(G):
        st, ed = g * gs, min((g + 1) * gs, K)
        m = mask[:, st:ed]
        mean[g], scale[g] = low_xnor_calibrate((W[:, st:ed] * m).astype(F32))
        q_high = high_quantize(W[:, st:ed], hscale, hzero, maxq)
        q_low = low_xnor_quantize(W[:, st:ed], mean[g], scale[g])
        out[:, st:ed] = (q_high * ~m).astype(F32) + (q_low * m).astype(F32)
    return dict(W_fq=out, mean=mean, scale=scale, hscale=hscale, hzero=hzero)


def gptq_blocks(W: np.ndarray, U: np.ndarray, mask: np.ndarray, hscale, hzero, maxq, mean, scale, gs: int, blocksize: int = 128):
    """The blocked column loop of LowHighGPT.fasterquant (gptq_pb/gptq.py:129-168) IN PLACE on W (fp32 [N, K]) for a given
    upper Cholesky factor U of H^-1, low mask and quantizer state; returns the per-row losses (:160,166).  Split out of
    ptq_gptq so that the loop can be checked against the reference with the reference's own U (golden G5 stores it)."""
    N, K = W.shape
    losses = np.zeros(N, F32)
    for c0 in range(0, K, blocksize):
        c1 = min(c0 + blocksize, K)
        W1 = W[:, c0:c1].copy()
        Q1 = np.zeros_like(W1)
        E1 = np.zeros_like(W1)
        L1 = np.zeros_like(W1)
        U1 = U[c0:c1, c0:c1]
        g = c0 // gs
        for i in range(c1 - c0):
            w = W1[:, i:i + 1]
            d = U1[i, i]
            q_high = high_quantize(w, hscale, hzero, maxq)
            q_low = low_xnor_quantize(w, mean[g], scale[g])
            m = mask[:, c0 + i:c0 + i + 1]
            q = (q_high * ~m).astype(F32) + (q_low * m).astype(F32)
            Q1[:, i] = q[:, 0]
            L1[:, i] = ((w - q) ** 2 / d ** 2)[:, 0]
            err = ((w - q) / d).astype(F32)
            W1[:, i:] -= err @ U1[i:i + 1, i:]
            E1[:, i] = err[:, 0]
        W[:, c0:c1] = Q1
        losses += L1.sum(1) / 2
        W[:, c1:] -= E1 @ U[c0:c1, c1:]
    return losses


def ptq_gptq(W: np.ndarray, H: np.ndarray, low_frac: float, metric: str = "magnitude",
             high_bit: int = 8, groupsize: int = -1, blocksize: int = 128, percdamp: float = 0.01):
    """Full LowHighGPT.fasterquant with GPTQ error feedback (gptq_pb/gptq.py:54-187).
    numpy float32 restatement of the blocked column loop (:129-168)."""
    W = W.astype(F32).copy()
    N, K = W.shape
    gs = K if groupsize == -1 else groupsize
    G = math.ceil(K / gs)
    hscale, hzero, maxq = high_calibrate(W, high_bit)              # :62-63
    U, dead = hinv_cholesky_upper(H, percdamp)                     # :67-81
    W[:, dead] = 0
    mask = ptq_low_mask(W, low_frac, metric, np.diag(U), groupsize)  # :83-99
    mean = np.zeros((G, N, 1), F32)
    scale = np.zeros((G, N, 1), F32)
    for g in range(G):
        st, ed = g * gs, min((g + 1) * gs, K)
        mean[g], scale[g] = low_xnor_calibrate((W[:, st:ed] * mask[:, st:ed]).astype(F32))
    losses = gptq_blocks(W, U, mask, hscale, hzero, maxq, mean, scale, gs, blocksize)
    return dict(W_fq=W, mask=mask, mean=mean, scale=scale, hscale=hscale, hzero=hzero,
                loss=float(losses.astype(np.float64).sum()), hinv_diag=np.diag(U).copy())


# --------------------------------------------------------------------------- #
# The arithmetic itself: F.linear (torch; call sites quant/quantizer.py:86,193,
# quant/outlier_quantizer.py:105 and every nn.Linear holding gptq.py:182 weights)
# --------------------------------------------------------------------------- #
def dense_linear(x: np.ndarray, W: np.ndarray, bias: np.ndarray | None = None) -> np.ndarray:
    """y = x @ W^T + b accumulated in float64 (the "truth" both torch's fp32-accumulate
    GEMM and the HIP kernel are compared against); result float64 [..., N]."""
    y = x.astype(np.float64).reshape(-1, x.shape[-1]) @ W.astype(np.float64).T
    if bias is not None:
        y = y + bias.astype(np.float64)
    return y.reshape(*x.shape[:-1], W.shape[0])


def pb_qat_forward(x, W_hat, mask, binary_scale, bias=None, outlier_scale=1.0):
    """BinaryXnorExceptOutliersLinear.forward (quant/outlier_quantizer.py:101-106)."""
    return dense_linear(x, binarize_except_outliers(W_hat, mask, binary_scale, outlier_scale), bias)


def binary_linear_forward(x, W, bias=None):
    """BinaryLinear.forward (quant/quantizer.py:84-86)."""
    return dense_linear(x, binary_linear_weight(W), bias)


def xnor_binary_linear_forward(x, W, bias=None):
    """XnorBinaryLinear.forward (quant/quantizer.py:191-193)."""
    return dense_linear(x, xnor_binary_linear_weight(W), bias)


# --------------------------------------------------------------------------- #
# One QAT training step (forward + straight-through backward), float64 "truth"
# --------------------------------------------------------------------------- #
def _bf16(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even to bfloat16, returned as float32 (the autocast cast in front of F.linear)"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def pb_qat_step(x, dy, W_hat, mask, bias=None, outlier_scale=1.0, train_outlier=False, gemm_bf16=False):
    """BinaryXnorExceptOutliersLinear in train() mode, forward and backward
    (quant/outlier_quantizer.py:83-106; STEBinary quant/quantizer.py:18-25).
      s      = mean |W[~mask]|, detached, a SCALAR (boolean indexing flattens)        :90-93
      w_sim  = where(mask, W*outlier_scale [detached unless train_outlier], sign(W)*s) :94-98
      y      = x w_sim^T + b                                                           :104-105
      dL/dx  = dy w_sim;  dL/dw_sim = dy^T x;  dL/db = sum dy
      dL/dW  = dL/dw_sim * (mask ? (train_outlier ? outlier_scale : 0) : s)   (STE: d sign(W)/dW := 1)
    gemm_bf16: what bf16 autocast (qat/run_qat.py:120) does -- x, w_sim, bias and dy are rounded to
    bf16 in front of the GEMMs, products accumulated wide.  Returns float64 arrays + s."""
    s = refresh_binary_scale(W_hat, mask)
    w_sim = binarize_except_outliers(W_hat, mask, s, outlier_scale).astype(np.float32)
    K, N = W_hat.shape[1], W_hat.shape[0]
    x2, dy2 = x.reshape(-1, K).astype(np.float32), dy.reshape(-1, N).astype(np.float32)
    b = None if bias is None else bias.astype(np.float32)
    if gemm_bf16:
        x2, dy2, w_sim = _bf16(x2), _bf16(dy2), _bf16(w_sim)
        b = None if b is None else _bf16(b)
    y = dense_linear(x2, w_sim, b).reshape(*x.shape[:-1], N)
    dx = (dy2.astype(np.float64) @ w_sim.astype(np.float64)).reshape(x.shape)
    g = dy2.astype(np.float64).T @ x2.astype(np.float64)
    coef = np.where(mask, float(outlier_scale) if train_outlier else 0.0, float(s.reshape(())))
    return dict(y=y, dx=dx, dW=g * coef, db=dy2.astype(np.float64).sum(0), binary_scale=s)


def ste_linear_step(x, dy, W, bias=None, xnor=False):
    """BinaryLinear / XnorBinaryLinear training step (quant/quantizer.py:84-86, 181-193).
    BinaryLinear: w = STE(W) -> dL/dW = dL/dw.  Xnor: c = W - rowmean(W); a = mean|c| detached;
    w = STE(c)*a -> dL/dc = dL/dw * a;  dL/dW = dL/dc - rowmean(dL/dc)."""
    K, N = W.shape[1], W.shape[0]
    w = xnor_binary_linear_weight(W) if xnor else binary_linear_weight(W)
    x2, dy2 = x.reshape(-1, K).astype(np.float64), dy.reshape(-1, N).astype(np.float64)
    y = dense_linear(x, w, bias)
    dx = (dy2 @ w.astype(np.float64)).reshape(x.shape)
    g = dy2.T @ x2
    if xnor:
        c = W.astype(np.float64) - W.astype(np.float64).mean(-1, keepdims=True)
        g = g * np.abs(c).mean(-1, keepdims=True)
        g = g - g.mean(-1, keepdims=True)
    return dict(y=y, dx=dx, dW=g, db=dy2.sum(0))


# --------------------------------------------------------------------------- #
# The reference's CPU path as it actually executes (torch), for bench.py's cpu_baseline leg
# --------------------------------------------------------------------------- #
def torch_dense_linear(x, W_fq, bias=None):
    """What gptq_pb/run.py / qat/eval_after_qat.py execute per layer on the host: stock
    nn.Linear.forward == F.linear over the dense fake-quant weight (gptq_pb/gptq.py:180-184)."""
    import torch.nn.functional as F
    return F.linear(x, W_fq, bias)


def torch_qat_forward_as_written(x, W_hat, outlier_mask, binary_scale, bias=None, outlier_scale=1.0):
    """BinaryXnorExceptOutliersLinear.forward as written (quant/outlier_quantizer.py:83-106):
    re-simulate the dense weight on EVERY call (mul, sign, mul, where), then F.linear."""
    import torch
    import torch.nn.functional as F
    scaled = W_hat * outlier_scale
    binary = torch.sign(W_hat) * binary_scale
    return F.linear(x, torch.where(outlier_mask, scaled, binary), bias)


# --------------------------------------------------------------------------- #
# Parity metric used throughout tests/ (SURVEY 8(c) "Tolerances")
# --------------------------------------------------------------------------- #
def parity_errors(y: np.ndarray, y_ref: np.ndarray):
    """(max|y-ref| / max|ref|,  max over elements of |y-ref| / (1e-3*rms(ref) + 1e-3*|ref|))."""
    y = y.astype(np.float64).reshape(-1)
    r = y_ref.astype(np.float64).reshape(-1)
    diff = np.abs(y - r)
    rel_max = float(diff.max() / max(np.abs(r).max(), 1e-30))
    rms = float(np.sqrt(np.mean(r * r)))
    allclose_ratio = float((diff / (1e-3 * rms + 1e-3 * np.abs(r) + 1e-30)).max())
    return rel_max, allclose_ratio
