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

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

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
            if high_scale is None:
                ss, sz = infer_code_grid(Wn, hi_l, lo_l, -1)
            else:
                ss = np.asarray(high_scale, np.float32).reshape(-1)
                sz = np.asarray(high_zero, np.float32).reshape(-1)
            shard = PBLinear(pack_dense(Wn[:, sl], hi_l, lo_l, ss, sz), bias if rank == 0 else None, W_fq.dtype)
        else:
            shard = PBLinear.from_dense(W_fq[:, sl], bias if rank == 0 else None,
                                        None if low_mask is None else low_mask[:, sl], groupsize,
                                        high_scale, high_zero)
    else:
        raise ValueError("mode must be 'n' or 'k'")
    return shard, (lo, hi)


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

    def __init__(self, shard: PBLinear, cols: tuple[int, int], group=None, input_is_sharded: bool = False):
        super().__init__()
        self.shard, self.cols, self.group, self.input_is_sharded = shard, cols, group, input_is_sharded

    def local_forward(self, x_local):
        """fp32 partial y of this rank's column slice (bias lives on rank 0 only)."""
        return pb_linear_forward(self.shard.packed, self.shard.pbl_bias, x_local, out_f32=True)

    def forward(self, x):
        xl = x if self.input_is_sharded else x[..., self.cols[0]:self.cols[1]]
        y = self.local_forward(xl).float().contiguous()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y.to(x.dtype)
