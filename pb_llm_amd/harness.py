"""Model-level harness: what gptq_pb/run.py:116-178 and
qat/eval_after_qat.py:12-24 do around the hot path, without datasets or checkpoints (none
are available offline): quantize every decoder Linear of a HF causal LM to a dense
fake-quant weight, swap in PBLinear modules, run forward / perplexity-style loops.

`quant_sequential_` is the counterpart of gptq_pb/run.py:quant_sequential on one MI355X: the whole model
stays resident in HBM (288 GB: no layer-by-layer host offload), every decoder Linear is quantised by the fused
GPTQ-PB pipeline (pb_llm_amd/ptq.py) and, optionally, packed on the spot.
`quantize_dense_` takes an external structure producer instead (tests pass the oracle's).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .quant import PBLinear, replace_linear_with_pb


def find_layers(module: nn.Module, name: str = "") -> dict:
    """gptq_pb/modelutils.py:8-16: every nn.Linear below `module`, by dotted name."""
    if isinstance(module, nn.Linear):
        return {name: module}
    res = {}
    for n, child in module.named_children():
        res.update(find_layers(child, name + "." + n if name else n))
    return res


@torch.no_grad()
def quantize_dense_(model: nn.Module, producer, skip=("lm_head",)) -> dict:
    """In place: W <- producer(name, W) -> dict(W_fq, low_mask, hscale, hzero) for every
    Linear (decoder layers only by default, like gptq_pb/run.py which leaves lm_head alone).
    Returns the per-layer side information for `to_pb_`."""
    side = {}
    for name, lin in find_layers(model).items():
        if any(s in name for s in skip):
            continue
        r = producer(name, lin.weight.data)
        lin.weight.data = r["W_fq"].to(lin.weight.dtype).to(lin.weight.device)
        side[name] = r
    return side


def to_pb_(model: nn.Module, side: dict | None = None, skip=("lm_head",)) -> nn.Module:
    """Swap every (fake-quant) Linear for a PBLinear.  With `side` the exact PTQ structure is
    used; without it the structure is re-inferred from the dense weights (flattened checkpoint)."""
    names = {id(m): n for n, m in model.named_modules()}

    def factory(lin: nn.Linear):
        s = (side or {}).get(names[id(lin)])
        if s is None:
            return PBLinear.from_dense(lin.weight.data, lin.bias).to(lin.weight.device)
        return PBLinear.from_dense(lin.weight.data, lin.bias, s.get("low_mask"), s.get("groupsize", -1),
                                   s.get("hscale"), s.get("hzero")).to(lin.weight.device)

    return replace_linear_with_pb(model, factory, skip)


def decoder_layers(model: nn.Module):
    """The list gptq_pb/run.py:49-63 picks per model family."""
    inner = getattr(model, "model", model)
    if hasattr(inner, "decoder") and hasattr(inner.decoder, "layers"):
        return inner.decoder.layers                      # OPT
    if hasattr(inner, "layers"):
        return inner.layers                              # LLaMA
    raise NotImplementedError("unknown decoder layout")


@torch.no_grad()
def quant_sequential_(model: nn.Module, calib_ids, low_frac: float, salient_metric: str = "magnitude", groupsize: int = -1,
                      high_bit: int = 8, disable_gptq: bool = False, percdamp: float = 0.01, pack: bool = False,
                      quant_only: str = "", log=None) -> dict:
    """gptq_pb/run.py:36-178 on the GPU.  calib_ids: iterable of token-id tensors [1, seqlen] (the dataloader's
    batch[0]).  Layer by layer: capture the layer's inputs, accumulate every Linear's Hessian through forward hooks
    (:148-156), quantise each Linear (:158-169), re-run the layer with the quantised weights to feed the next one
    (:171-172).  pack=True swaps each quantised Linear for a PBLinear right away.  Returns {name: error}."""
    from .ptq import LowHighGPTQ
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("quant_sequential_ needs the model on the GPU: there is no host path")
    layers = decoder_layers(model)
    use_cache = getattr(model.config, "use_cache", False)
    model.config.use_cache = False
    captured = []

    def catch(module, args, kwargs):                     # the reference's Catcher (:72-83)
        captured.append((args, kwargs))
        raise ValueError

    handle = layers[0].register_forward_pre_hook(catch, with_kwargs=True)
    for ids in calib_ids:
        try:
            model(ids.to(dev))
        except ValueError:
            pass
    handle.remove()
    inps = [a[0] if a else k["hidden_states"] for a, k in captured]
    errors = {}
    for i, layer in enumerate(layers):
        subset = {n: m for n, m in find_layers(layer).items() if quant_only in n}
        gpts = {n: LowHighGPTQ(m, salient_metric, groupsize, high_bit, disable_gptq) for n, m in subset.items()}
        handles = [subset[n].register_forward_hook(lambda _, inp, out, n=n: gpts[n].add_batch(inp[0].data, out.data))
                   for n in gpts]

        def run(j, hidden):
            args, kwargs = captured[j]
            if args:
                out = layer(hidden, *args[1:], **kwargs)
            else:
                out = layer(**{**kwargs, "hidden_states": hidden})
            return out[0] if isinstance(out, tuple) else out

        for j, h in enumerate(inps):
            run(j, h)
        for h in handles:
            h.remove()
        for n, g in gpts.items():
            info = g.fasterquant(low_frac, percdamp=percdamp)
            errors[f"{i}.{n}"] = info["error"]
            if log:
                log(f"{i} {n} error {info['error']:.4f}")
            if pack:
                parent = layer
                *path, leaf = n.split(".")
                for p_ in path:
                    parent = getattr(parent, p_)
                setattr(parent, leaf, g.to_pb().to(dev))
            g.free()
        inps = [run(j, h) for j, h in enumerate(inps)]
    model.config.use_cache = use_cache
    return errors


class _FusedMember(nn.Module):
    """Stands where q_proj / k_proj / v_proj (or gate_proj / up_proj) stood.  WHICHEVER member is called first with an
    activation launches the whole group once (runtime.FusedGemv) and the others pick their slice up when they are called
    with the SAME tensor object (HF attention / MLP modules pass one hidden_states to all projections).  The group keeps
    a reference to that tensor until the next launch: an identity match can therefore not be a recycled device address
    (the caching allocator hands the next token's activation the address of the previous one), and the call order of the
    members does not matter.  A member called with another tensor -- a view, a copy, the next token -- launches again;
    launches that served no other member are counted (`_FusedGroup.solo_launches`) and reported once, because a model
    that never shares the tensor pays the whole group per projection.  More than 4 rows, or an fp32 activation: each
    member runs its own PBLinear (matrix-core / GEMM regime), bit-identical to the unfused model.  bf16 activations (HF LLaMA
    checkpoints, qat/run_qat.py:120) run the same fused launch with the conversion inside the kernel (round 5)."""

    def __init__(self, group: "_FusedGroup", index: int, own: PBLinear):
        super().__init__()
        self.own = own                      # registered: state_dict / .to() keep working
        self._group, self._index = [group], index          # (list: keep the group out of the module tree)
        self.in_features, self.out_features = own.in_features, own.out_features
        self.global_name = own.global_name

    def forward(self, x):
        g = self._group[0]
        rows = x.numel() // x.shape[-1]
        if not x.is_cuda or (torch.is_grad_enabled() and x.requires_grad):
            return self.own(x)
        fused_ok = rows <= 4 and (x.dtype == torch.float16 or (x.dtype == torch.bfloat16 and g.fused.bf16_ok))
        # batched decode (5 .. MERGE_MAX_ROWS rows, any activation dtype): ONE call of the row-wise concatenation of the members
        # (packing.concat_rows: a record is 16 rows with everything it needs inside it) instead of one small-batch launch + reduce
        # per projection
        merged_ok = not fused_ok and g.merge_batched and 4 < rows <= g.MERGE_MAX_ROWS
        if not (fused_ok or merged_ok):
            return self.own(x)
        if g.x_ref is x and g.x_version == x._version and self._index in g.pending:     # each launch serves each member once
            g.pending.discard(self._index)
            g.served += 1
            return g.outs[self._index].reshape(*x.shape[:-1], self.out_features)
        g.launch(x, rows, self._index, merged=merged_ok)
        return g.outs[self._index].reshape(*x.shape[:-1], self.out_features)


class _FusedGroup:
    """One fused launch (runtime.FusedGemv) over the member PBLinears.  The launch descriptors snapshot the members' blobs
    (device pointers, LDS sizing): they are rebuilt when a member's blob moved or was rewritten (.to(), load_state_dict)."""
    SOLO_WARN_AFTER = 32
    MERGE_MAX_ROWS = 64            # the small-batch kernel's range: beyond it a merged layer has nothing on its members (same tiles, same rounds)

    def __init__(self, mods: list[PBLinear], merge_batched: bool = False):
        self.mods = mods
        self.x_ref, self.x_version, self.outs, self.pending = None, -1, None, set()
        self.launches = self.served = self.solo_launches = self.merged_launches = 0
        self._warned = False
        self.merged = None          # the members as ONE PBLinear (rows concatenated), built on the first batched call
        self.merge_batched = merge_batched and all(m.out_features % 16 == 0 for m in mods[:-1]) and \
            len({(m.packed.K, m.packed.G, m.packed.flags) for m in mods}) == 1
        self._build()

    def _stamp(self):
        return tuple((m.pbl_blob.data_ptr(), m.pbl_blob._version, None if m.pbl_bias is None else m.pbl_bias.data_ptr())
                     for m in self.mods)

    def _build(self):
        from .runtime import FusedGemv
        dev = self.mods[0].pbl_blob.device
        self.fused = FusedGemv([m.packed for m in self.mods], [m.pbl_bias for m in self.mods], dev)
        self.merged = None
        self.stamp = self._stamp()

    def _merged(self) -> PBLinear:
        if self.merged is None:
            from .packing import concat_rows
            biases = [m.pbl_bias for m in self.mods]
            bias = None
            if any(b is not None for b in biases):
                dev = self.mods[0].pbl_blob.device
                bias = torch.cat([b if b is not None else torch.zeros(m.out_features, device=dev) for b, m in zip(biases, self.mods)])
            self.merged = PBLinear(concat_rows([m.packed for m in self.mods]), bias, self.mods[0].weight_dtype)
            import logging
            logging.getLogger("pb_llm_amd").info("fuse_decode_(merge_batched=True): merged layer of %d projections holds %d more packed bytes "
                                                 "(+ its GEMM image on the first batched call)", len(self.mods), self.merged.packed.nbytes)
            self.offs = [0]
            for m in self.mods:
                self.offs.append(self.offs[-1] + m.out_features)
        return self.merged

    def launch(self, x, rows, index, merged: bool = False):
        if self.pending and self.launches and len(self.pending) == len(self.mods) - 1:
            self.solo_launches += 1          # the previous launch served nobody but its caller
            if self.solo_launches == self.SOLO_WARN_AFTER and not self._warned:
                import warnings
                self._warned = True
                warnings.warn("pb_llm_amd.harness: fused projection group keeps launching for single members -- the model does "
                              "not pass one tensor object to its q/k/v (gate/up) projections; fuse_decode_ gains nothing here")
        if self._stamp() != self.stamp:
            self._build()
        if merged:
            y = self._merged()(x.reshape(rows, x.shape[-1]))
            self.outs = [y[:, self.offs[i]:self.offs[i + 1]] for i in range(len(self.mods))]
            self.merged_launches += 1
        else:
            self.outs = self.fused(x.reshape(rows, x.shape[-1]).contiguous())
        self.x_ref, self.x_version = x, x._version
        self.pending = set(range(len(self.mods))) - {index}
        self.launches += 1


FUSE_SETS = (("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"))     # LLaMA naming (HF)


def fuse_decode_(model: nn.Module, sets=FUSE_SETS, merge_batched: bool = False) -> int:
    """Decode-time fusion: in every module that holds all the projections of a set as PBLinears with one
    in_features (HF LlamaAttention: q/k/v_proj; LlamaMLP: gate/up_proj), replace them by members of one fused launch.
    7 launches per decoder layer become 4.  Returns the number of groups created.  The caller this serves is the token
    loop of gptq_pb/eval_ppl_utils.py:55-64 / qat/eval_after_qat.py:11-33 at batch 1.
    merge_batched (round 5; OPT-IN since round 6, ADVICE r5: the default silently held blob + image + merged blob + merged image): a
    decode BATCH (5 - 64 rows) runs each group as ONE call of the members' row-wise concatenation (packing.concat_rows; built on the
    first such call; costs the members' packed bytes once more plus its GEMM image -- logged once per group at INFO level on the
    "pb_llm_amd" logger) instead of one small-batch launch + reduce per projection (llama-7b decoder layer at 16 rows: 94.0 -> 78.5
    us).  A decode-only phase at batch <= 4 never reads a GEMM image: `drop_gemm_images_` releases them."""
    n = 0
    for mod in list(model.modules()):
        for names in sets:
            subs = [getattr(mod, nm, None) for nm in names]
            if not all(isinstance(m_, PBLinear) for m_ in subs):
                continue
            if len({m_.in_features for m_ in subs}) != 1 or not subs[0].pbl_blob.is_cuda:
                continue
            grp = _FusedGroup(subs, merge_batched)
            for i, (nm, m_) in enumerate(zip(names, subs)):
                setattr(mod, nm, _FusedMember(grp, i, m_))
            n += 1
    return n


def build_gemm_images_(model: nn.Module, release_blobs: bool = False) -> tuple[int, int]:
    """Build and keep the GEMM image of every packed fp16-checkpoint linear of `model` NOW (quant._kept_image), instead of on its
    first prefill call: afterwards calls with 5 - 64 rows run the small-batch kernel over the image under the default
    quant.SMALL_BATCH_IMAGE = "auto" -- also inside a hipGraph capture, where an image cannot be built (the build reads two words
    back).  Costs the images' memory on top of the blobs (1.7 - 2.4 x the blob's bytes) -- unless release_blobs (round 6): then the
    image becomes each layer's ONLY device copy (PBLinear.release_blob_: the blob moves to host memory, state_dict() still holds it;
    every row count multiplies from the image) -- what an evaluation run wants (qat/eval_after_qat.py:11-33, gptq_pb/eval_ppl_utils.py:
    55-64 call the layers with whole 2048-token windows only): a llama-7b-shaped model at 5 % hessian salients holds 4.2 GB instead
    of 6.3.  Returns (layers with an image, image bytes); `release_blobs_` / `restore_blobs_` do the second step on their own."""
    from . import quant as Q
    n = nbytes = 0
    for m in model.modules():
        if isinstance(m, Q.PBLinear) and m._image_only is not None:
            n += 1
            nbytes += m._image_only[0].data.numel()
        elif isinstance(m, Q.PBLinear) and m.weight_dtype == torch.float16 and m.pbl_blob.is_cuda and Q.fused_gemm_ok(m.packed):
            img = Q._kept_image(m.packed)
            if img is not None:
                n += 1
                nbytes += img.data.numel()
    if release_blobs:
        release_blobs_(model)
    return n, nbytes


def release_blobs_(model: nn.Module, pin: bool = False) -> int:
    """PBLinear.release_blob_ on every packed linear of `model` that has a GEMM image (fp16-checkpoint layers; others keep their
    blob); returns the device bytes released.  Not on a model whose projections were fused for decode (fuse_decode_: the fused
    launches read the blobs) -- un-fused layers released here are simply skipped by a later fuse_decode_."""
    from . import quant as Q
    if any(isinstance(m, _FusedMember) for m in model.modules()):
        raise RuntimeError("release_blobs_: the model's projections are fused for decode (fuse_decode_), whose launches read the blobs")
    n = 0
    for m in model.modules():
        if isinstance(m, Q.PBLinear) and m._image_only is None and m.pbl_blob.is_cuda and m.weight_dtype == torch.float16 \
                and Q.fused_gemm_ok(m.packed) and Q._kept_image(m.packed) is not None:
            n += m.release_blob_(pin)
    torch.cuda.empty_cache()
    return n


def restore_blobs_(model: nn.Module) -> int:
    """the blobs of every image-only linear back on the device (decode through the GEMV again); returns the layers restored"""
    from . import quant as Q
    n = 0
    for m in model.modules():
        if isinstance(m, Q.PBLinear) and m._image_only is not None:
            m.restore_blob_()
            n += 1
    return n


def drop_gemm_images_(model: nn.Module) -> int:
    """Release every kept GEMM image (and kept salient list) of `model`'s packed linears -- the memory knob next to
    build_gemm_images_: with the round-5 defaults every fp16-exact linear gets an image on its first call with 5 rows or more
    (1.7 - 2.4 x the blob's bytes; 4.2 GB for a 7B model at 5 % hessian salients), which a decode-only phase at batch <= 4 never
    reads.  The next call that wants an image rebuilds it.  Not while a captured graph that uses the images may still be replayed.
    Returns the bytes released."""
    from . import quant as Q
    n = 0
    for m in model.modules():
        if isinstance(m, Q.PBLinear):
            p = m.packed
            kept = p.__dict__.pop("_gemm_image", None)
            if kept is not None and kept[1] is not None:
                n += kept[1].data.numel()
            lst = p.__dict__.pop("_gemm_list", None)
            if lst is not None and lst[1] is not None:
                n += lst[1].numel()
    return n


class GraphedForward:
    """One forward of `model` on a fixed input shape captured in a hipGraph (the C ABI launches are asynchronous and
    allocation free; torch's caching allocator serves the temporaries from the graph's private pool).  replay(ids) copies
    the new token ids into the static input and replays: per-token host cost is one graph launch instead of hundreds of
    Python-side kernel launches."""

    def __init__(self, model: nn.Module, example_ids: torch.Tensor, warmup: int = 2):
        self.model = model
        self.static_ids = example_ids.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):
                model(self.static_ids, use_cache=False)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_logits = model(self.static_ids, use_cache=False).logits

    def replay(self, ids: torch.Tensor) -> torch.Tensor:
        self.static_ids.copy_(ids)
        self.graph.replay()
        return self.static_logits


@torch.no_grad()
def perplexity(model: nn.Module, input_ids: torch.Tensor, seqlen: int) -> float:
    """The loop of gptq_pb/eval_ppl_utils.py:55-86 / evaluate.py:126-156 on pre-tokenised ids
    [1, n*seqlen]: mean NLL over non-overlapping windows -> exp."""
    n = input_ids.numel() // seqlen
    nll = 0.0
    for i in range(n):
        batch = input_ids[:, i * seqlen:(i + 1) * seqlen]
        logits = model(batch).logits.float()
        loss = nn.functional.cross_entropy(logits[0, :-1], batch[0, 1:], reduction="sum")
        nll += float(loss)
    return float(torch.exp(torch.tensor(nll / (n * (seqlen - 1)))))
