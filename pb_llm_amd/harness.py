"""Model-level harness: what gptq_pb/run.py:116-178 and
qat/eval_after_qat.py:12-24 do around the hot path, without datasets or checkpoints (none
are available offline): quantize every decoder Linear of a HF causal LM to a dense
fake-quant weight, swap in PBLinear modules, run forward / perplexity-style loops.

`quantize_dense_` stands in for `quant_sequential` with the RTN branch (`--disable_gptq`):
it needs a structure producer; tests pass the oracle's, a user passes gptq_pb's outputs.
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
