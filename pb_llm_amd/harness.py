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
