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

    _FIELDS = ("blob", "N", "K", "P", "G", "NRB", "flags", "max_nch", "max_nexc", "nnz", "nexc")

    def __getstate__(self):
        """state = the dataclass fields; what a forward hangs on the object (the cached ctypes descriptor, a kept GEMM image or
        salient list with its stream event) is derived data: copy.deepcopy / pickle of a layer that has already run must not trip
        over a ctypes pointer or drag an image along -- the copy rebuilds them on first use"""
        return {f: getattr(self, f) for f in self._FIELDS}

    def __setstate__(self, state):
        self.__dict__.update(state)

    @property
    def nbytes(self) -> int:
        return int(self.blob.numel())

    def to(self, device) -> "PackedWeight":
        return PackedWeight(self.blob.to(device), self.N, self.K, self.P, self.G, self.NRB, self.flags,
                            self.max_nch, self.max_nexc, self.nnz, self.nexc)

    def layer_struct(self, bias: torch.Tensor | None = None) -> _lib.PblLayer:
        """The C descriptor of the layer (borrowed pointers).  Cached per (blob address, bias address): an eager decode calls
        this 224 times per token, and building a ctypes structure costs as much as the launch it describes."""
        bp, wp = (bias.data_ptr() if bias is not None else 0), self.blob.data_ptr()
        hit = self.__dict__.get("_struct")
        if hit is not None and hit[0] == wp and hit[1] == bp:
            return hit[2]
        if wp % 16:
            raise _lib.PblError("blob is not 16-byte aligned")
        st = _lib.PblLayer(wp, bp or None, self.N, self.K, self.P, self.G, self.NRB, self.flags, self.max_nch, self.max_nexc)
        self.__dict__["_struct"] = (wp, bp, st)
        return st

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


def concat_rows(parts: "list[PackedWeight]") -> PackedWeight:
    """ONE packed layer whose output rows are the rows of `parts` in order (q | k | v, gate | up: projections that read the same
    activation; the reference calls them as separate nn.Linear modules, gptq_pb/eval_ppl_utils.py:55-64 through the HF attention /
    MLP blocks).  A record is 16 output rows with everything it needs inside it, so this is byte surgery, not re-packing: the
    records are copied behind one another and the record table is rebuilt with shifted offsets -- on whatever device the blobs live
    on.  Needs equal in_features, column groups and flags, and every part but the last a whole number of 16-row records.  The
    result unpacks to the row-wise concatenation of the parts' matrices, bit for bit."""
    if not parts:
        raise ValueError("nothing to concatenate")
    p0 = parts[0]
    for p in parts:
        if (p.K, p.G, p.P, p.flags) != (p0.K, p0.G, p0.P, p0.flags) or p.blob.device != p0.blob.device:
            raise _lib.PblError("concat_rows: parts must share in_features, column groups, flags and device")
    for p in parts[:-1]:
        if p.N % 16:
            raise _lib.PblError("concat_rows: every part but the last must be a whole number of 16-row records")
    NRB = sum(p.NRB for p in parts)
    N = sum(p.N for p in parts)
    _check_limits(N, p0.K, p0.G)
    rec0 = (80 + 16 * (NRB + 1) + 127) & ~127
    dev = p0.blob.device
    infos, recs, cur = [], [], rec0
    for p in parts:
        r0 = (80 + 16 * (p.NRB + 1) + 127) & ~127                        # where this part's records start
        info = p.blob[80:80 + 16 * p.NRB].view(torch.int32).reshape(p.NRB, 4).clone()
        info[:, 0] += (cur - r0) // 16                                   # off16: record offsets in units of 16 bytes (both multiples of 128)
        infos.append(info)
        recs.append(p.blob[r0:])
        cur += p.nbytes - r0
    if cur // 16 >= 1 << 31:
        raise _lib.PblError("concat_rows: the concatenated blob exceeds the record table's 32-bit offsets")
    infos.append(torch.tensor([[cur // 16, 0, 0, 0]], dtype=torch.int32, device=dev))
    hdr = _lib.PblBlobHeader.from_buffer_copy(bytes(p0.blob[:80].cpu().numpy()))
    hdr.N, hdr.NRB = N, NRB
    hdr.max_nch, hdr.max_nexc = max(p.max_nch for p in parts), max(p.max_nexc for p in parts)
    hdr.nnz, hdr.nexc, hdr.blob_bytes = sum(p.nnz for p in parts), sum(p.nexc for p in parts), cur
    head = torch.zeros(rec0, dtype=torch.uint8, device=dev)
    head[:80] = torch.frombuffer(bytearray(bytes(hdr)), dtype=torch.uint8).to(dev)
    table = torch.cat(infos, 0).contiguous().view(torch.uint8).reshape(-1)
    head[80:80 + table.numel()] = table
    blob = torch.cat([head] + recs)
    return PackedWeight(blob, N, p0.K, p0.P, p0.G, NRB, p0.flags, hdr.max_nch, hdr.max_nexc, hdr.nnz, hdr.nexc)


MAX_IN_FEATURES = 32767          # 16-bit column indices in the salient list (include/pbl.h, "Limits")
MAX_OUT_FEATURES = 1 << 24


def _check_limits(N: int, K: int, G: int) -> None:
    """The format's limits, spelled out before libpbl answers with a bare PBL_ERR_UNSUPPORTED."""
    if K > MAX_IN_FEATURES:
        raise _lib.PblError(f"in_features = {K}: the PBL1 format indexes columns with 16 bits (at most {MAX_IN_FEATURES}); "
                            "shard the layer along K (parallel.shard_linear(..., mode='k')) and sum the partial outputs")
    if N > MAX_OUT_FEATURES:
        raise _lib.PblError(f"out_features = {N}: at most {MAX_OUT_FEATURES} rows per packed layer")
    if G > 1 and (K % G or (K // G) % 128):
        raise _lib.PblError(f"groupsize {K / G:g}: column groups must divide in_features and be multiples of 128 "
                            "(power-of-two sizes for the kernels: 128, 256, 512, ...)")


def pack_dense(W, hi, lo, sscale=None, szero=None, sal_mask=None, sal_f16: bool = False) -> PackedWeight:
    """Pack a dense simulated weight.  W [N,K]; hi, lo [N,G] (the two values the
    binarized weights of each row/group take); sscale, szero [N] (salient value =
    sscale*(q - szero), HighQuantizer's form, gptq_pb/high_quant.py:6-8) or None.
    Exact for any input: values on neither level nor the code grid are stored as
    fp32 exceptions.  sal_f16: W came from an fp16 checkpoint, i.e. salient values are
    fl16(sscale*(q-szero)) (PBL_FLAG_SAL_F16)."""
    W = _f32(W)
    N, K = W.shape
    hi = _f32(hi).reshape(N, -1)
    lo = _f32(lo).reshape(N, -1)
    G = hi.shape[1]
    if lo.shape != hi.shape:
        raise ValueError("hi / lo shape mismatch")
    _check_limits(N, K, G)
    ss = _f32(sscale).reshape(N) if sscale is not None else None
    sz = _f32(szero).reshape(N) if szero is not None else None
    sm = None
    if sal_mask is not None:
        sm = np.ascontiguousarray(
            sal_mask.detach().cpu().numpy() if isinstance(sal_mask, torch.Tensor) else sal_mask).astype(np.uint8)
    L = _lib.lib()
    ptr = lambda a: a.ctypes.data if a is not None else None  # noqa: E731
    size = C.c_size_t(0)
    flags = _lib.PBL_FLAG_SAL_F16 if sal_f16 else 0
    _lib.check(L.pbl_pack_dense_f32(ptr(W), N, K, G, ptr(hi), ptr(lo), ptr(ss), ptr(sz), ptr(sm), flags,
                                    None, 0, C.byref(size)), "pack(size)")
    blob = torch.empty(size.value, dtype=torch.uint8)
    _lib.check(L.pbl_pack_dense_f32(ptr(W), N, K, G, ptr(hi), ptr(lo), ptr(ss), ptr(sz), ptr(sm), flags,
                                    blob.data_ptr(), size.value, C.byref(size)), "pack")
    return PackedWeight.from_blob(blob)


def pack_dense_dev(W: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, sscale: torch.Tensor | None = None,
                   szero: torch.Tensor | None = None, sal_mask: torch.Tensor | None = None, sal_f16: bool = False) -> PackedWeight:
    """pack_dense for tensors that already live on the GPU: the blob is built there (csrc/pbl_pack.hip, one wavefront per
    record) and is byte-identical to the host packer's.  No host round trip of the weights; one small device -> host copy
    (the blob's size and header totals)."""
    if not W.is_cuda:
        raise _lib.PblError("pack_dense_dev needs GPU tensors (use pack_dense on the host)")
    dev = W.device
    W = W.detach().to(torch.float32).contiguous()
    N, K = W.shape
    f32 = lambda t: None if t is None else torch.as_tensor(t, device=dev).detach().to(torch.float32).contiguous()   # noqa: E731
    hi, lo = f32(hi).reshape(N, -1), f32(lo).reshape(N, -1)
    G = hi.shape[1]
    _check_limits(N, K, G)
    if lo.shape != hi.shape:
        raise ValueError("hi / lo shape mismatch")
    ss, sz = f32(sscale), f32(szero)
    sm = None if sal_mask is None else torch.as_tensor(sal_mask, device=dev).reshape(N, K).to(torch.uint8).contiguous()
    L = _lib.lib()
    NRB, CW = (N + 15) // 16, _lib.PBL_PACK_COUNT_WORDS
    counts = torch.empty(NRB, CW, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ptr = lambda t: None if t is None else t.data_ptr()      # noqa: E731
    flags = _lib.PBL_FLAG_SAL_F16 if sal_f16 else 0
    _lib.check(L.pbl_pack_dev_count(W.data_ptr(), N, K, G, hi.data_ptr(), lo.data_ptr(), ptr(ss), ptr(sz), ptr(sm), flags,
                                    counts.data_ptr(), st), "pack_dev_count")
    c = counts.to(torch.int64)
    rec0 = (80 + 16 * (NRB + 1) + 127) & ~127
    rec_off = torch.empty(NRB + 1, dtype=torch.int64, device=dev)
    rec_off[0] = rec0
    rec_off[1:] = rec0 + torch.cumsum(c[:, 0], 0)
    tot = torch.stack([rec_off[-1], (c[:, 1] + c[:, 2]).max(), c[:, 3].max(), c[:, 4].sum(), c[:, 3].sum(), c[:, 5].max()]).tolist()
    total, max_nch, max_nexc, nnz, nexc, bad = (int(v) for v in tot)
    if bad:
        raise _lib.PblError("layer exceeds the packed format's limits (chunks per record / tail chunks per row)")
    blob = torch.empty(total, dtype=torch.uint8, device=dev)
    _lib.check(L.pbl_pack_dev_write(W.data_ptr(), N, K, G, hi.data_ptr(), lo.data_ptr(), ptr(ss), ptr(sz), ptr(sm), flags,
                                    counts.data_ptr(), rec_off.data_ptr(), total, max_nch, max_nexc, nnz, nexc, blob.data_ptr(), st),
               "pack_dev_write")
    fl = (_lib.PBL_FLAG_HAS_GROUPS if G > 1 else 0) | flags | _lib.PBL_FLAG_TAIL_REPEAT | _lib.PBL_FLAG_SLABS
    return PackedWeight(blob, N, K, (K + 511) // 512, G, NRB, fl, max_nch, max_nexc, nnz, nexc)


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


def _best_scale(v: np.ndarray, n: np.ndarray, cands: np.ndarray, sal_f16: bool):
    """Among fp32 candidate scales, the one reproducing most values v == fl(c * n) exactly."""
    best = (-1, np.float32(cands[0]))
    nf = n.astype(np.float32)
    for c in cands:
        rec = (np.float32(c) * nf).astype(np.float32)
        if sal_f16:
            rec = rec.astype(np.float16).astype(np.float32)
        hits = int(np.count_nonzero(rec == v))
        if hits > best[0]:
            best = (hits, np.float32(c))
    return best[1]


def _spectral_scale(v64: np.ndarray) -> float | None:
    """Grid pitch of values known to be (noisy) integer multiples of an unknown step: the
    LARGEST s whose integrality score  mean cos(2 pi v / s)  is near the maximum (sub-multiples
    s/2, s/3 score as well, hence "largest").  Used when the quick estimates fail, e.g. after
    GPTQ error feedback has moved the row extremes off codes 0 / 255."""
    span = float(v64[-1] - v64[0])
    if span <= 0:
        return None
    sub = v64 if v64.size <= 96 else v64[np.linspace(0, v64.size - 1, 96).astype(int)]
    s_grid = np.exp(np.arange(np.log(span / 255.0 * 0.98), np.log(span / 4.0), 4e-4))
    score = np.cos(2.0 * np.pi * sub[None, :] / s_grid[:, None]).mean(1)
    good = np.nonzero(score >= 0.9 * score.max())[0]
    return float(s_grid[good[-1]]) if good.size else None


def infer_code_grid(W: np.ndarray, hi: np.ndarray, lo: np.ndarray, groupsize: int = -1, sal_f16: bool = False):
    """Per-row affine grid (sscale, szero) of the values that are on neither level, i.e.
    HighQuantizer's scale*(q-zero) (gptq_pb/high_quant.py:6-8) re-discovered from a dense
    checkpoint.  Estimate: the row's extreme off-level values are codes 0 and 255 (the row's
    min / max weight are always salient), refined by least squares and an exact-match search
    over neighbouring fp32 scales (through the fp16 round trip when sal_f16).  Values that
    stay off the inferred grid become exceptions in the packer, so a poor guess costs bytes,
    never correctness."""
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
            sz[r] = 2.0 if v[0] < 0 else 0.0     # value = ss*(1 - sz)
            continue
        v64 = v.astype(np.float64)
        gaps = np.diff(v64)
        gmin = float(gaps[gaps > 0].min())
        # coarse scales: full range / 255 (both extreme codes present: magnitude saliency) and the
        # smallest gap or a fraction of it (adjacent codes need not both occur)
        coarse = [(v64[-1] - v64[0]) / 255.0, gmin, gmin / 2.0, gmin / 3.0, None]
        best = None
        for s0 in coarse:
            if s0 is None:                          # last resort, only if nothing fitted well
                if best is not None and best[0] >= 0.98 * v.size:
                    break
                s0 = _spectral_scale(v64)
                if s0 is None:
                    break
            s1, ok = s0, True
            for lim in (4, 16, 64, 1 << 30):       # progressive least squares, small |n| first:
                n = np.rint(v64 / s1)               # the coarse estimate only has to be right to
                sel = (np.abs(n) <= lim) & (n != 0)  # within 0.5/lim for the codes it is fitted on
                if sel.any():
                    s1 = float((v64[sel] * n[sel]).sum() / (n[sel] * n[sel]).sum())
            n = np.rint(v64 / s1)
            if n.max() - n.min() > 255 or not np.any(n):
                continue
            width = 6e-5 if sal_f16 else 4e-7
            cands = np.unique((s1 * (1.0 + np.linspace(-width, width, 241))).astype(np.float32))
            c = _best_scale(v, n, cands, sal_f16)
            rec = (c * n.astype(np.float32)).astype(np.float32)
            if sal_f16:
                rec = rec.astype(np.float16).astype(np.float32)
            hits = int(np.count_nonzero(rec == v))
            if best is None or hits > best[0]:
                # any integer zero point with all codes in [0,255] reproduces the same values
                best = (hits, c, np.float32(-n.min()))
        if best is not None:
            ss[r], sz[r] = best[1], best[2]
    return ss, sz
