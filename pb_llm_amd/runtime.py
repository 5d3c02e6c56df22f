"""Launch helpers above the C ABI: a group of independent PB linears in ONE kernel
launch (fused QKV / gate+up at decode time; the L-layer stream benchmark of
SURVEY.md 8(d)).  Device-resident descriptor tables are built once and reused, so
a launch is a single pbl_gemv_f16_grouped call and is hipGraph-capturable.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .packing import PackedWeight


class GroupedGemv:
    """y_l = x_l @ W_l^T (+ b_l) for l in 0..L-1, all in one launch.  M <= 4 tokens."""

    def __init__(self, packed: list[PackedWeight], biases: list[torch.Tensor | None] | None = None,
                 M: int = 1, device="cuda", shared_x: bool = False, out_f32: bool = False):
        if not packed:
            raise ValueError("empty group")
        if not 1 <= M <= _lib.PBL_MAX_TOKENS_PER_LAUNCH:
            raise ValueError("grouped launch handles 1..4 tokens")
        self.device = torch.device(device)
        self.M = M
        self.packed = [p if p.blob.device == self.device else p.to(self.device) for p in packed]
        biases = biases or [None] * len(packed)
        self.biases = [b.detach().float().to(self.device) if b is not None else None for b in biases]
        if shared_x and len({p.K for p in self.packed}) != 1:
            raise ValueError("shared_x needs equal in_features")
        Kmax = max(p.K for p in self.packed)
        if shared_x:
            xs = torch.zeros(M, Kmax, dtype=torch.float16, device=self.device)
            self.x = [xs] * len(self.packed)
        else:
            self.x = [torch.zeros(M, p.K, dtype=torch.float16, device=self.device) for p in self.packed]
        # one contiguous output buffer (a single all-reduce covers the whole group when sharded)
        self.out_f32 = out_f32
        self.y_all = torch.empty(sum(M * p.N for p in self.packed), dtype=torch.float32 if out_f32 else torch.float16,
                                 device=self.device)
        offs = np.cumsum([0] + [M * p.N for p in self.packed])
        self.y = [self.y_all[offs[i]:offs[i + 1]].view(M, p.N) for i, p in enumerate(self.packed)]
        structs = (_lib.PblLayer * len(self.packed))(*[p.layer_struct(b) for p, b in zip(self.packed, self.biases)])
        raw = np.frombuffer(bytes(structs), dtype=np.uint8).copy()
        self._layers_dev = torch.from_numpy(raw).to(self.device)
        self._x_ptrs = torch.tensor([t.data_ptr() for t in self.x], dtype=torch.int64, device=self.device)
        self._y_ptrs = torch.tensor([t.data_ptr() for t in self.y], dtype=torch.int64, device=self.device)
        self.max_NRB = max(p.NRB for p in self.packed)
        self.max_K = Kmax
        self.max_nch = max(p.max_nch for p in self.packed)
        self.max_nexc = max(p.max_nexc for p in self.packed)
        # bit 0: column-group layers present (the launch then runs the column-group kernel), bit 1: fp16-checkpoint layers present
        self.any_groups = int(any(p.G > 1 for p in self.packed)) | (2 * int(any(p.flags & _lib.PBL_FLAG_SAL_F16 for p in self.packed)))

    def algorithmic_bytes(self) -> int:
        return sum(p.algorithmic_bytes(self.M, b is not None) for p, b in zip(self.packed, self.biases))

    def packed_bytes(self) -> int:
        return sum(p.nbytes for p in self.packed)

    def launch(self, stream: torch.cuda.Stream | None = None) -> list[torch.Tensor]:
        st = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        _lib.check(_lib.lib().pbl_gemv_f16_grouped(
            self._layers_dev.data_ptr(), self._x_ptrs.data_ptr(), self._y_ptrs.data_ptr(), len(self.packed),
            self.M, self.max_NRB, self.max_K, self.max_nch, self.max_nexc, self.any_groups, int(self.out_f32), st),
            "grouped gemv")
        return self.y


class FusedGemv:
    """Decode-time fusion of projections that read the same activation (q/k/v; gate/up): ONE launch, one output tensor.
    __call__(x [M <= 4, K] fp16) -> list of [M, N_l] views of a fresh [M, sum N_l] tensor.  No pointer table refers to x
    or y (pbl_gemv_f16_fused), so every call may use new tensors and the launch can be captured in a hipGraph."""

    def __init__(self, packed: list[PackedWeight], biases: list[torch.Tensor | None] | None = None, device="cuda"):
        if not packed:
            raise ValueError("empty group")
        if len({p.K for p in packed}) != 1:
            raise ValueError("fused projections share their input: equal in_features")
        self.device = torch.device(device)
        self.packed = [p if p.blob.device == self.device else p.to(self.device) for p in packed]
        biases = biases or [None] * len(packed)
        self.biases = [b.detach().float().to(self.device) if b is not None else None for b in biases]
        self.K = self.packed[0].K
        self.Ns = [p.N for p in self.packed]
        self.offs = np.cumsum([0] + self.Ns)
        structs = (_lib.PblLayer * len(self.packed))(*[p.layer_struct(b) for p, b in zip(self.packed, self.biases)])
        self._layers_dev = torch.from_numpy(np.frombuffer(bytes(structs), dtype=np.uint8).copy()).to(self.device)
        self._yoff_dev = torch.tensor(self.offs[:-1].tolist(), dtype=torch.int64, device=self.device)
        # up to PBL_FUSED_INLINE_MAX members: descriptors and offsets go into the kernel arguments (pbl_gemv_f16_fused_host)
        self._inline = len(self.packed) <= _lib.PBL_FUSED_INLINE_MAX
        self._structs_host = structs
        self._yoff_host = (C.c_uint64 * len(self.packed))(*[int(o) for o in self.offs[:-1]])
        self.max_NRB = max(p.NRB for p in self.packed)
        self.max_nch = max(p.max_nch for p in self.packed)
        # bit 0: column-group layers present (the launch then runs the column-group kernel), bit 1: fp16-checkpoint layers present
        self.flags = int(any(p.G > 1 for p in self.packed)) | 2 * int(any(p.flags & _lib.PBL_FLAG_SAL_F16 for p in self.packed))
        self.bf16_ok = self._inline and not (self.flags & 1)

    def __call__(self, x2: torch.Tensor, out_f32: bool = False) -> list[torch.Tensor]:
        M = x2.shape[0]
        if x2.dtype not in (torch.float16, torch.bfloat16) or not x2.is_contiguous() or x2.shape[1] != self.K or not 1 <= M <= _lib.PBL_MAX_TOKENS_PER_LAUNCH:
            raise _lib.PblError("FusedGemv: x must be a contiguous fp16 / bf16 [M <= 4, K] tensor")
        total = int(self.offs[-1])
        y = torch.empty(M, total, dtype=torch.float32 if out_f32 else x2.dtype, device=x2.device)
        st = torch.cuda.current_stream(x2.device).cuda_stream
        L = _lib.lib()
        if x2.dtype == torch.bfloat16:
            # bf16 activations converted IN the launch, bf16 result (pbl_gemv_bf16_fused_host; group-free members, at most
            # PBL_FUSED_INLINE_MAX of them: what q/k/v and gate/up are)
            if not self.bf16_ok:
                raise _lib.PblError("FusedGemv: bf16 activations need group-free members and at most 4 of them")
            _lib.check(L.pbl_gemv_bf16_fused_host(C.addressof(self._structs_host), C.addressof(self._yoff_host), x2.data_ptr(), y.data_ptr(),
                                                  len(self.packed), M, total, self.max_NRB, self.K, self.max_nch, self.flags,
                                                  int(out_f32), st), "fused gemv (bf16)")
        elif self._inline:
            _lib.check(L.pbl_gemv_f16_fused_host(C.addressof(self._structs_host), C.addressof(self._yoff_host), x2.data_ptr(), y.data_ptr(),
                                                 len(self.packed), M, total, self.max_NRB, self.K, self.max_nch, self.flags,
                                                 int(out_f32), st), "fused gemv")
        else:
            _lib.check(L.pbl_gemv_f16_fused(self._layers_dev.data_ptr(), self._yoff_dev.data_ptr(), x2.data_ptr(), y.data_ptr(),
                                            len(self.packed), M, total, self.max_NRB, self.K, self.max_nch, self.flags,
                                            int(out_f32), st), "fused gemv")
        return [y[:, int(self.offs[i]):int(self.offs[i + 1])] for i in range(len(self.packed))]

    def algorithmic_bytes(self, M: int = 1) -> int:
        # x is read once for the group, not once per layer
        return sum(p.algorithmic_bytes(M, b is not None) for p, b in zip(self.packed, self.biases)) - 2 * M * self.K * (len(self.packed) - 1)
