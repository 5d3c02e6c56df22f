"""Tensor-parallel sharding of a PB linear across GPUs.

The reference has no distributed code at all.  A PB linear shards like
any matrix product:

* N-split ("column-parallel"): each rank owns N/P output rows -> its own PBL1 blob of
  those rows; x is replicated; NO exchange (outputs are concatenated, optionally
  all-gathered).
* K-split ("row-parallel", what BASELINE's "row-sharded ... all-reduce" denotes): each
  rank owns a slice of the input columns of every row: its columns of the sign plane,
  the salient entries whose column falls in the slice, and -- because hi/lo multiply
  sums of x -- the same per-row levels.  Partial y is summed by ONE all-reduce of [M,N]
  (fp32 partials; 16 KB at N=4096, M=1).

One process per GPU; `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Shards are cut on the DENSE simulated weight and packed per rank,
so every shard is an ordinary PBLinear and uses the same kernel.
"""
from __future__ import annotations

import ctypes as C
import itertools
import weakref

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from .quant import PBLinear, pb_linear_forward

ROW_ALIGN = 16     # records are 16 rows
COL_ALIGN = 128    # column groups are multiples of 128; keeps shard boundaries group-aligned


def split_points(total: int, world: int, align: int) -> list[int]:
    """world+1 boundaries, multiples of `align` (except the last), as even as possible.
    E.g. llama-7b down_proj K=11008 = 86*128 over 8 ranks -> 11,11,11,11,11,11,10,10 blocks."""
    blocks = (total + align - 1) // align
    base, extra = divmod(blocks, world)
    pts = [0]
    for r in range(world):
        pts.append(min(total, pts[-1] + (base + (1 if r < extra else 0)) * align))
    pts[-1] = total
    return pts


def shard_linear(W_fq: torch.Tensor, bias, low_mask, mode: str, rank: int, world: int, groupsize: int = -1,
                 high_scale=None, high_zero=None) -> tuple[PBLinear, tuple[int, int]]:
    """Build this rank's PBLinear shard from the dense fake-quant weight (+ optional
    PTQ side information).  Returns (shard, (lo, hi)) with the row/column range owned."""
    N, K = W_fq.shape
    if mode == "n":
        lo, hi = split_points(N, world, ROW_ALIGN)[rank:rank + 2]
        sl = slice(lo, hi)
        hs = None if high_scale is None else torch.as_tensor(high_scale).reshape(-1)[sl]
        hz = None if high_zero is None else torch.as_tensor(high_zero).reshape(-1)[sl]
        shard = PBLinear.from_dense(W_fq[sl], None if bias is None else bias[sl],
                                    None if low_mask is None else low_mask[sl], groupsize, hs, hz)
    elif mode == "k":
        if groupsize != -1 and groupsize % COL_ALIGN:
            raise ValueError("groupsize must be a multiple of 128")
        lo, hi = split_points(K, world, COL_ALIGN if groupsize == -1 else groupsize)[rank:rank + 2]
        sl = slice(lo, hi)
        if low_mask is None and groupsize == -1:
            # levels are per ROW: infer them on the full row so every shard agrees
            from .packing import infer_levels, infer_code_grid, pack_dense
            Wn = W_fq.detach().cpu().float().numpy()
            hi_l, lo_l = infer_levels(Wn, -1, None)
            f16 = W_fq.dtype == torch.float16      # fp16 checkpoint: salients are fl16(scale*(q-zero)), like from_dense
            if high_scale is None:
                ss, sz = infer_code_grid(Wn, hi_l, lo_l, -1, sal_f16=f16)
            else:
                ss = np.asarray(high_scale, np.float32).reshape(-1)
                sz = np.asarray(high_zero, np.float32).reshape(-1)
            shard = PBLinear(pack_dense(Wn[:, sl], hi_l, lo_l, ss, sz, sal_f16=f16), bias if rank == 0 else None, W_fq.dtype)
        else:
            shard = PBLinear.from_dense(W_fq[:, sl], bias if rank == 0 else None,
                                        None if low_mask is None else low_mask[:, sl], groupsize,
                                        high_scale, high_zero)
    else:
        raise ValueError("mode must be 'n' or 'k'")
    return shard, (lo, hi)


class P2PAllReduce:
    """One-shot fp32 sum all-reduce over peer-mapped buffers (libpbl, csrc/pbl_comm.hip): the native
    collective for the small messages of K-split decode, where RCCL's all-reduce is latency bound.  One process per GPU.
    Every rank allocates one communication buffer through libpbl, the 64-byte IPC handles travel through
    torch.distributed (any backend: gloo in the one-GPU test, RCCL on a node), every rank maps the others' buffers.
    all_reduce_(t) then is a single kernel launch on the current stream: push to 7 peers over xGMI, flag, wait, local
    sum in rank order (bit-identical on every rank).  The call number lives in the buffer (pbl_p2p_allreduce_f32_dev), so
    the launch is hipGraph-capturable and replayable.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC).

    `P2PAllReduce.shared(device, group)` hands every K-split layer of a model the SAME communicator: one buffer of
    2 * world * max_numel floats per rank (the slots are strided by the world size), allocated at the first all-reduce with
    the largest message any layer registered -- a 7B model with 64 K-split layers used to allocate 64 buffers of 33 MB."""

    _shared: dict = {}          # (device index, serial) -> (weakref to the process group | None for the default group, communicator)
    _serial = itertools.count()

    def __init__(self, max_numel: int, device, group=None, lazy: bool = False):
        self.group, self.device = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 16:
            raise ValueError("at most 16 ranks")
        self.max_numel = int(max_numel)
        self._own, self._opened, self._ptrs = None, [], None
        # diagnostics (round 6; bench.py --gpus N "tp_phases"): a list here makes fused_linear_ / all_reduce_ record HIP events around
        # their launches -- (kind, start, end) with kind in {"push_gemv", "reduce", "p2p_allreduce"} -- so that the first run on real
        # xGMI says WHERE a step's time went.  None (default): nothing is recorded.
        self.phase_events = None
        if not lazy:
            self._allocate()

    @classmethod
    def shared(cls, device, group=None, max_numel: int = 0) -> "P2PAllReduce":
        """the communicator of (device, group), created on first use; max_numel only ever grows until the first all-reduce"""
        # keyed on the group OBJECT (held weakly): an id() can be recycled after a group is garbage collected and would then
        # hand a communicator mapped for other ranks to a new group
        dev_index = torch.device(device).index
        for k, (gref, c) in list(cls._shared.items()):
            g = gref() if gref is not None else None
            if gref is not None and g is None:
                del cls._shared[k]                      # the group is gone: its communicator can never be asked for again
                continue
            if k[0] == dev_index and g is group:
                c.reserve(max_numel)
                return c
        c = cls(max_numel, device, group, lazy=True)
        cls._shared[(dev_index, next(cls._serial))] = (weakref.ref(group) if group is not None else None, c)
        return c

    def reserve(self, numel: int):
        """grow the capacity while the buffers are not allocated yet; afterwards the capacity is frozen and larger messages
        take the caller's RCCL path (PBLinearKSplit.forward checks max_numel) -- a layer built after warm-up must not fail"""
        if numel > self.max_numel and self._own is None:
            self.max_numel = int(numel)

    def _allocate(self):
        """collective: every rank allocates, exports, maps (at the first all_reduce_ of a lazily created communicator:
        all ranks run the same model, so they get here together).  A failure on ANY rank (no IPC support, a mapping that is
        refused) is agreed over the group before anybody launches: every rank then releases what it holds and raises PblError, so
        that the caller can fall back to RCCL on all ranks together instead of one rank raising while its peers wait in a barrier."""
        L = _lib.lib()
        group = self.group
        err = None
        with torch.cuda.device(self.device):
            own = C.c_void_p()
            handle = (C.c_ubyte * 64)()
            try:
                _lib.check(L.pbl_comm_alloc(L.pbl_p2p_buffer_bytes_world(self.max_numel, self.world), C.byref(own)), "comm_alloc")
                self._own = own.value
                _lib.check(L.pbl_ipc_export(self._own, handle), "ipc_export")
            except _lib.PblError as e:
                err = e
            handles = [None] * self.world
            dist.all_gather_object(handles, (err is None, bytes(handle)), group=group)
            self._ptrs = (C.c_void_p * self.world)()
            self._opened = []
            if err is None and all(ok for ok, _ in handles):
                try:
                    for r, (_, h) in enumerate(handles):
                        if r == self.rank:
                            self._ptrs[r] = self._own
                            continue
                        ptr = C.c_void_p()
                        buf = (C.c_ubyte * 64).from_buffer_copy(h)
                        _lib.check(L.pbl_ipc_open(buf, C.byref(ptr)), f"ipc_open(rank {r})")
                        self._ptrs[r] = ptr.value
                        self._opened.append(ptr.value)
                except _lib.PblError as e:
                    err = e
            elif err is None:
                err = _lib.PblError("P2PAllReduce: a peer could not allocate / export its communication buffer")
        # nobody launches before everybody has mapped everybody -- and everybody learns whether everybody did
        if not agree_min(int(err is None), group, self.device):
            for p in self._opened:
                L.pbl_ipc_close(p)
            if self._own:
                L.pbl_comm_free(self._own)
            self._own, self._opened, self._ptrs = None, [], None
            raise _lib.PblError(f"P2PAllReduce: peer mapping failed on at least one rank ({err or 'on a peer'}); use collective='rccl'")

    def all_reduce_(self, t: torch.Tensor, out_f16: torch.Tensor | None = None) -> torch.Tensor:
        """t (fp32, contiguous) <- sum over ranks, in place; out_f16 (optional, same numel, fp16) also receives the sum
        rounded to fp16 by the same kernel"""
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            raise _lib.PblError("P2PAllReduce: contiguous fp32 tensor on the communicator's device")
        if out_f16 is not None and (out_f16.dtype != torch.float16 or not out_f16.is_contiguous() or out_f16.numel() != t.numel()):
            raise _lib.PblError("P2PAllReduce: out_f16 must be a contiguous fp16 tensor of the same size")
        if self._own is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.PblError("P2PAllReduce: run one all-reduce eagerly before capturing (the buffers are mapped on first use)")
            self._allocate()
        if t.numel() > self.max_numel:
            raise _lib.PblError(f"P2PAllReduce sized for {self.max_numel} elements, got {t.numel()}")
        st = torch.cuda.current_stream(self.device).cuda_stream
        ev = self._phase_start()
        _lib.check(_lib.lib().pbl_p2p_allreduce_f32_dev(self._ptrs, self.rank, self.world, t.data_ptr(),
                                                        out_f16.data_ptr() if out_f16 is not None else None, t.numel(),
                                                        self.max_numel, st), "p2p_allreduce")
        self._phase_end("p2p_allreduce", ev)
        return t

    def _phase_start(self):
        if self.phase_events is None or torch.cuda.is_current_stream_capturing():
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.device))
        return e

    def _phase_end(self, kind, e0):
        if e0 is None:
            return None
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream(self.device))
        self.phase_events.append((kind, e0, e1))
        return e1

    @staticmethod
    def selftest_pairs(device, numel: int = 4096, iters: int = 20, group=None) -> dict:
        """One-shot all-reduce latency of `numel` fp32 (16 KB by default: one decode token of a 4096-wide layer) for every PAIR of
        ranks, over a two-rank communicator of its own: {"i-j": microseconds} on every rank (max over the two members).  Collective
        over `group` (every rank takes part in creating every pair's group).  What to read it for: on a node each pair is one xGMI
        link, and a slow or absent link shows here before it shows as a slow step."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ranks = list(range(world)) if group is None else dist.get_process_group_ranks(group)
        out = {}
        for i in range(world):
            for j in range(i + 1, world):
                pg = dist.new_group([ranks[i], ranks[j]])               # (collective over the default group: every rank calls it)
                us = -1.0
                if rank in (i, j):
                    try:
                        c = P2PAllReduce(numel, device, pg)
                        t = torch.ones(numel, dtype=torch.float32, device=device)
                        for _ in range(3):
                            c.all_reduce_(t)
                        torch.cuda.synchronize(device)
                        dist.barrier(group=pg)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(iters):
                            c.all_reduce_(t)
                        e1.record()
                        torch.cuda.synchronize(device)
                        c.check()
                        us = e0.elapsed_time(e1) * 1e3 / iters
                        c.close()
                    except _lib.PblError:
                        us = -1.0
                res = [None] * world
                dist.all_gather_object(res, us, group=group)
                out[f"{i}-{j}"] = round(max(res[i], res[j]), 2) if min(res[i], res[j]) >= 0 else None
        return out

    def fused_linear_(self, packed, bias_f32, x2: torch.Tensor, out: torch.Tensor) -> bool:
        """K-split layer with the push fused into the GEMV's epilogue (pbl_linear_f16_push) + the reduce (pbl_p2p_reduce_f32_dev):
        x2 [M, K_shard] fp16 contiguous, out [M, N] fp16 or fp32 receives the all-reduced result.  False: this call is not one GEMV
        pass (more than 4 tokens, column groups) -- the caller runs the unfused pair; nothing was launched.  EVERY rank must make the
        same choice for a call (PBLinearKSplit.push_max_tokens is agreed over the group for that reason); a caller of its own keeps M
        at or under the minimum of pbl_linear_push_max_tokens over the ranks."""
        if self._own is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.PblError("P2PAllReduce: run one all-reduce eagerly before capturing (the buffers are mapped on first use)")
            self._allocate()
        M = x2.shape[0]
        if M * packed.N > self.max_numel:
            return False
        L = _lib.lib()
        layer = packed.layer_struct(bias_f32)
        st = torch.cuda.current_stream(self.device).cuda_stream
        ev = self._phase_start()
        rc = L.pbl_linear_f16_push(C.byref(layer), x2.data_ptr(), M, self._ptrs, self.rank, self.world, self.max_numel, st)
        if rc == _lib.PBL_ERR_UNSUPPORTED:
            return False
        _lib.check(rc, "linear_f16_push")
        ev = self._phase_end("push_gemv", ev)
        f32 = out.dtype == torch.float32
        _lib.check(L.pbl_p2p_reduce_f32_dev(self._ptrs, self.rank, self.world, out.data_ptr() if f32 else None,
                                            None if f32 else out.data_ptr(), M * packed.N, self.max_numel, packed.NRB, st), "p2p_reduce")
        self._phase_end("reduce", ev)
        return True

    def check(self) -> None:
        """synchronous: raises if any wait timed out since the last check() (a peer died or never launched; the affected outputs
        were overwritten with NaN by the kernel).  The status word is cleared when it is reported (round 6): one transient stall
        poisons the calls up to this check, not every later one."""
        if self._own is not None and _lib.lib().pbl_p2p_check(self._own) != 0:
            raise _lib.PblError("P2PAllReduce: a peer's flag did not arrive within the bounded wait")

    def close(self):
        L = _lib.lib()
        if getattr(self, "_own", None):
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)   # no peer may still be writing into a buffer that is about to go
            for p in self._opened:
                L.pbl_ipc_close(p)
            L.pbl_comm_free(self._own)
            self._own, self._opened = None, []
        for k, v in list(P2PAllReduce._shared.items()):
            if v[1] is self:
                del P2PAllReduce._shared[k]


def agree_min(value: int, group=None, device=None) -> int:
    """the minimum of `value` over the ranks of `group` (collective).  The tensor lives where the group's backend wants it:
    on the GPU for RCCL ("nccl"), on the host for gloo."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(value)
    on_gpu = "nccl" in str(dist.get_backend(group))
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


class PBLinearNSplit(nn.Module):
    """Output rows split across ranks; no reduction.  gather_output=True all-gathers y."""

    def __init__(self, shard: PBLinear, rows: tuple[int, int], out_features: int, group=None,
                 gather_output: bool = True):
        super().__init__()
        self.shard, self.rows, self.out_features = shard, rows, out_features
        self.group, self.gather_output = group, gather_output

    def local_forward(self, x):
        return self.shard(x)

    def forward(self, x):
        y = self.local_forward(x)
        if not self.gather_output:
            return y
        world = dist.get_world_size(self.group)
        pts = split_points(self.out_features, world, ROW_ALIGN)
        width = max(pts[r + 1] - pts[r] for r in range(world))
        pad = torch.zeros(*y.shape[:-1], width, dtype=y.dtype, device=y.device)
        pad[..., : y.shape[-1]] = y
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([parts[r][..., : pts[r + 1] - pts[r]] for r in range(world)], -1)


class PBLinearKSplit(nn.Module):
    """Input columns split across ranks; partial outputs summed by one all-reduce (fp32)."""

    def __init__(self, shard: PBLinear, cols: tuple[int, int], group=None, input_is_sharded: bool = False,
                 collective: str = "rccl", max_tokens: int = 64):
        """collective: "rccl" = torch.distributed all_reduce (RCCL over xGMI on a node); "p2p" = libpbl's one-shot
        peer-to-peer all-reduce (P2PAllReduce) for up to `max_tokens` rows, RCCL above that."""
        super().__init__()
        self.shard, self.cols, self.group, self.input_is_sharded = shard, cols, group, input_is_sharded
        if collective not in ("rccl", "p2p"):
            raise ValueError("collective must be 'rccl' or 'p2p'")
        self.comm, self.max_tokens = None, max_tokens
        # a timed-out peer wait poisons the output with NaN and sets the communicator's status word; the word is polled
        # (synchronously) every `check_every` eager forwards, so a dead peer surfaces as an exception, not only as NaN logits
        self.check_every, self._calls = 256, 0
        self.fuse_push = True       # <= push_max_tokens fp16 rows: pbl_linear_f16_push + pbl_p2p_reduce_f32_dev instead of GEMV + all-reduce
        self.push_max_tokens = 0
        if collective == "p2p":
            # ONE communicator per (device, group) for all K-split layers, sized to the largest message
            self.comm = P2PAllReduce.shared(shard.pbl_blob.device, group, max_tokens * shard.out_features)
            # The fused pair and GEMV + all-reduce wait on DIFFERENT words of the peers' buffers (record counters / flags): every
            # rank must take the same one for a given M.  How many tokens one push pass takes depends on the rank's OWN shard (its
            # fullest record sizes the LDS: pbl_linear_push_max_tokens), so the ranks agree on the minimum here, once
            # (collective: every rank constructs its K-split layers in the same order).
            self.push_max_tokens = agree_min(_lib.lib().pbl_linear_push_max_tokens(C.byref(shard.packed.layer_struct(None))),
                                             group, shard.pbl_blob.device)

    def local_forward(self, x_local):
        """fp32 partial y of this rank's column slice (bias lives on rank 0 only)."""
        return pb_linear_forward(self.shard.packed, self.shard.pbl_bias, x_local, out_f32=True)

    def forward(self, x):
        xl = x if self.input_is_sharded else x[..., self.cols[0]:self.cols[1]]
        if self.comm is not None and self.fuse_push and x.dtype == torch.float16 and not (torch.is_grad_enabled() and x.requires_grad):
            # decode: ONE GEMV pass whose epilogue pushes the partial to every rank + the reduce (round 4): no all-reduce launch
            # that first reads the partial back from local HBM
            x2 = xl.reshape(-1, xl.shape[-1]).contiguous()
            if 0 < x2.shape[0] <= self.push_max_tokens:
                self._calls += 1
                if self.check_every and self._calls % self.check_every == 0 and not torch.cuda.is_current_stream_capturing():
                    self.comm.check()
                out = torch.empty(x2.shape[0], self.shard.out_features, dtype=torch.float16, device=x.device)
                if self.comm.fused_linear_(self.shard.packed, self.shard.pbl_bias, x2, out):
                    return out.reshape(*x.shape[:-1], self.shard.out_features)
        y = self.local_forward(xl)               # fp32, contiguous: the kernels' own output
        if self.comm is not None and y.numel() <= self.comm.max_numel:
            self._calls += 1
            if self.check_every and self._calls % self.check_every == 0 and not torch.cuda.is_current_stream_capturing():
                self.comm.check()
            if x.dtype == torch.float16:         # the all-reduce kernel writes the fp16 result itself: no cast launch
                out = torch.empty(y.shape, dtype=torch.float16, device=y.device)
                self.comm.all_reduce_(y, out)
                return out
            self.comm.all_reduce_(y)
        else:
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y.to(x.dtype)
