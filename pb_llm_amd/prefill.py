"""Prefill pipeline for the GEMM regime (more than 32 rows of x: the reference's perplexity loops,
gptq_pb/eval_ppl_utils.py:55-64 and evaluate.py:126-145, call every decoder Linear with seq 2048 rows, layer after layer).

In that regime a packed layer needs a per-layer preparation that does not depend on x before its GEMM can run:
  library backend  pbl_unpack_dev: the dense fp16 weight in a scratch buffer (HBM bound, 13-40 us on the llama-7b shapes)
  fused backend    pbl_gemm_prepare: the layer's salient list (10-30 us)
The GEMM itself is matrix-core bound (60-190 us).  Run back to back on one stream the preparation is pure overhead (a
llama-7b-shaped forward at seq 2048: 45.2 ms packed vs 39.4 ms dense, profiles/r03_gemm.md).  The pipeline issues the
preparation of the NEXT layer on a second, high-priority HIP stream while the current layer's GEMM runs on the caller's stream:
two scratch slots (2 x the largest layer: 2 x 90 MB for llama-7b -- nothing on a 288 GB part), events in both directions,
no host synchronisation.  Which layer comes next is learned from the call order (seeded with module registration order); a
wrong guess costs one wasted preparation and the layer prepares itself in line, so the result never depends on the guess.

The dense weight still never outlives two layers; the packed blobs stay the only resident copy of the model.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from . import quant as Q


class PrefillPipeline:
    """with PrefillPipeline(model): model(ids)      -- or  pipe = PrefillPipeline(model).install() ... pipe.remove()"""

    def __init__(self, model: nn.Module):
        self.mods = [m for m in model.modules() if isinstance(m, Q.PBLinear)]
        if not self.mods:
            raise ValueError("no PBLinear modules in the model")
        dev = self.mods[0].pbl_blob.device
        if dev.type != "cuda" or any(m.pbl_blob.device != dev for m in self.mods):
            raise _lib.PblError("the prefill pipeline needs every PBLinear on one GPU")
        self.device = dev
        self.index = {id(m.pbl_blob): i for i, m in enumerate(self.mods)}
        self.next_of = {i: (i + 1) % len(self.mods) for i in range(len(self.mods))}       # seed: registration order
        self.prev = None
        self.side = torch.cuda.Stream(dev, priority=-1)
        self.bufs = {}           # kind -> [tensor, tensor] (flat uint8)
        self.holds = {}          # kind -> [key | None, key | None]
        self.ready = {}          # kind -> [event, event]: the slot's content is complete (recorded where it was written)
        self.released = {}       # kind -> [event, event]: the last GEMM reading the slot has been issued (caller's stream)
        self.last = {}           # kind -> slot the current / last GEMM reads
        self.stats = {"hits": 0, "inline": 0, "prefetches": 0, "bypassed": 0}

    # ---- installation ----------------------------------------------------------------------------------------------------
    def install(self):
        Q.PREFILL = self
        return self

    def remove(self):
        if Q.PREFILL is self:
            Q.PREFILL = None
        torch.cuda.synchronize(self.device)      # nothing of ours is in flight when the slots go back to the allocator
        self.bufs.clear()

    def __enter__(self):
        return self.install()

    def __exit__(self, *exc):
        self.remove()

    # ---- slots -------------------------------------------------------------------------------------------------------------
    def _need(self, kind):
        if kind == "dense":
            return max(m.out_features * m.in_features * 2 for m in self.mods)
        L = _lib.lib()
        return max(int(L.pbl_gemm_list_bytes(C.byref(m.packed.layer_struct(None)))) for m in self.mods)

    def _slots(self, kind):
        if kind not in self.bufs:
            n = (self._need(kind) + 255) & ~255
            self.bufs[kind] = [torch.empty(n, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.holds[kind] = [None, None]
            self.ready[kind] = [torch.cuda.Event(), torch.cuda.Event()]
            self.released[kind] = [torch.cuda.Event(), torch.cuda.Event()]
            self.last[kind] = 0
        return self.bufs[kind]

    @staticmethod
    def _key(packed):
        return (packed.blob.data_ptr(), packed.blob._version, packed.N, packed.K)

    def _prepare(self, kind, packed, slot):
        """issue the preparation of `packed` into slot `slot` on the CURRENT stream"""
        buf = self.bufs[kind][slot]
        if kind == "dense":
            Q.unpack_on_device(packed, torch.float16, out=buf[:packed.N * packed.K * 2].view(torch.float16).view(packed.N, packed.K))
        else:
            layer = packed.layer_struct(None)
            _lib.check(_lib.lib().pbl_gemm_prepare(C.byref(layer), buf.data_ptr(), buf.numel(),
                                                   torch.cuda.current_stream(self.device).cuda_stream), "gemm_prepare")

    def acquire(self, packed, kind):
        """The layer's prepared scratch (kind "dense": the [N, K] fp16 weight; "list": the flat salient-list workspace), valid on
        the current stream; None: the layer is not one of the pipeline's (the caller prepares as without a pipeline).  Call
        release(kind) after the GEMM that reads it has been issued."""
        i = self.index.get(id(packed.blob))
        if i is None or torch.cuda.is_current_stream_capturing():
            self.stats["bypassed"] += 1
            return None
        if kind == "list" and not _lib.lib().pbl_gemm_list_bytes(C.byref(packed.layer_struct(None))):
            return None
        self._slots(kind)
        holds, ready, released = self.holds[kind], self.ready[kind], self.released[kind]
        main = torch.cuda.current_stream(self.device)
        key = self._key(packed)
        if self.prev is not None and self.prev != i:
            self.next_of[self.prev] = i                      # learn the call order
        self.prev = i
        s = holds.index(key) if key in holds else None
        if s is not None:
            main.wait_event(ready[s])
            self.stats["hits"] += 1
        else:
            s = self.last[kind] ^ 1
            main.wait_event(ready[s])                         # a (mispredicted) preparation may still be writing the slot
            self._prepare(kind, packed, s)
            holds[s] = key
            ready[s].record(main)
            self.stats["inline"] += 1
        self.last[kind] = s
        # the next layer's preparation, into the other slot, on the side stream
        nxt = self.mods[self.next_of[i]].packed
        nkey = self._key(nxt)
        if nkey not in holds and (kind == "dense" or _lib.lib().pbl_gemm_list_bytes(C.byref(nxt.layer_struct(None)))):
            o = s ^ 1
            self.side.wait_event(released[o])                 # the GEMM that last read the slot
            self.side.wait_event(ready[o])                    # ... and whoever wrote it last (in line, on the caller's stream)
            with torch.cuda.stream(self.side):
                self._prepare(kind, nxt, o)
                ready[o].record(self.side)
            holds[o] = nkey
            self.stats["prefetches"] += 1
        buf = self.bufs[kind][s]
        if kind == "dense":
            return buf[:packed.N * packed.K * 2].view(torch.float16).view(packed.N, packed.K)
        return buf

    def release(self, kind):
        self.released[kind][self.last[kind]].record(torch.cuda.current_stream(self.device))
