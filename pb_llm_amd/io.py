"""On-disk format for packed models.

Mirrors the reference's only PB-native checkpoint scheme, utils.py:65-124
(`save_bnn` / `load_bnn`, dead code there): a directory with
    meta.json    {module name -> class name, shape, dtype, ...}
    weights.pth  torch.save dict {name + "_blob": uint8 PBL1 blob, name + "_bias": tensor|None}
where the reference stores {name + "_weight": dense fp16}.  A blob is self-describing
(include/pbl.h header), so meta.json is informational plus what the module needs to rebuild.
Also: the per-layer low-mask files gptq_pb dumps (gptq_pb/gptq.py:108-114), which
BinaryXnorExceptOutliersLinearHessian loads (quant/outlier_quantizer.py:126-143).
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

from .packing import PackedWeight
from .quant import BinaryInterface, PBLinear

_DTYPES = {"float16": torch.float16, "float32": torch.float32, "bfloat16": torch.bfloat16}


def get_pb_meta(model: nn.Module) -> dict:
    meta = {}
    for name, m in model.named_modules():
        if isinstance(m, BinaryInterface):
            p = m.packed if isinstance(m, PBLinear) else m._packed_on(next(m.parameters()).device)
            meta[name] = {"class": m.__class__.__name__, "N": p.N, "K": p.K, "G": p.G, "flags": p.flags,
                          "nnz": p.nnz, "nexc": p.nexc, "bytes": p.nbytes,
                          "dtype": str(m.weight_dtype if isinstance(m, PBLinear) else m.weight.dtype).replace("torch.", ""),
                          "global_name": getattr(m, "global_name", None)}
    return meta


def save_pb(model: nn.Module, save_path: str) -> dict:
    """Write meta.json + weights.pth for every BinaryInterface module of `model`."""
    os.makedirs(save_path, exist_ok=True)
    meta = get_pb_meta(model)
    weights = {}
    for name, m in model.named_modules():
        if name in meta:
            p = m.packed if isinstance(m, PBLinear) else m._packed
            weights[name + "_blob"] = p.blob.detach().cpu()
            b = m.bias
            weights[name + "_bias"] = None if b is None else b.detach().float().cpu()
    with open(os.path.join(save_path, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    torch.save(weights, os.path.join(save_path, "weights.pth"))
    return meta


def load_pb(model: nn.Module, load_path: str) -> nn.Module:
    """Replace every module of `model` named in meta.json (nn.Linear or an earlier PB module)
    by a PBLinear over the stored blob -- attribute replacement as in utils.py:103-123."""
    with open(os.path.join(load_path, "meta.json")) as f:
        meta = json.load(f)
    weights = torch.load(os.path.join(load_path, "weights.pth"), weights_only=True)
    modules = dict(model.named_modules())
    for name, info in meta.items():
        if name not in modules:
            raise KeyError(f"checkpoint module {name!r} not found in the model")
        old = modules[name]
        packed = PackedWeight.from_blob(weights[name + "_blob"])
        if isinstance(old, nn.Linear) and (old.in_features, old.out_features) != (packed.K, packed.N):
            raise ValueError(f"{name}: shape mismatch")
        new = PBLinear(packed, weights[name + "_bias"], _DTYPES.get(info.get("dtype", "float16"), torch.float16))
        new.global_name = info.get("global_name")
        ind = name.rfind(".")
        father = modules[""] if ind == -1 else modules[name[:ind]]
        setattr(father, name[ind + 1:], new)
    return model


def save_bnn(model: nn.Module, save_path: str) -> dict:
    """The REFERENCE's layout (utils.py:87-94), for interchange: meta.json {module name -> class name} and weights.pth
    {name + "_weight": the module's weight parameter as fp16 (get_save_weight_dict, quant/quantizer.py:70-72), name + "_bias"}."""
    os.makedirs(save_path, exist_ok=True)
    meta, weights = {}, {}
    for name, m in model.named_modules():
        if isinstance(m, BinaryInterface):
            meta[name] = m.__class__.__name__
            for k, v in m.get_save_weight_dict().items():
                weights[name + "_" + k] = v.detach().cpu() if isinstance(v, torch.Tensor) else v
    with open(os.path.join(save_path, "meta.json"), "w") as f:
        json.dump(meta, f)
    torch.save(weights, os.path.join(save_path, "weights.pth"))
    return meta


def _load_weights_file(path: str, allow_pickle: bool):
    """weights.pth of a save_bnn directory.  The reference stores biases as nn.Parameter objects (utils.py:78-83), which the
    restricted unpickler only takes with the Parameter rebuild helpers allow-listed; anything else in the file needs
    allow_pickle=True (a full unpickle runs arbitrary code from a foreign directory, as the reference's torch.load does)."""
    import torch._utils as tu
    safe = [nn.Parameter, tu._rebuild_parameter]
    if hasattr(tu, "_rebuild_parameter_with_state"):
        safe.append(tu._rebuild_parameter_with_state)
    try:
        with torch.serialization.safe_globals(safe):
            return torch.load(path, weights_only=True)
    except Exception as e:  # noqa: BLE001 -- whatever the restricted unpickler raises, the decision is the caller's
        if not allow_pickle:
            raise ValueError(f"{path}: not loadable with the restricted unpickler ({type(e).__name__}: {e}); "
                             "pass allow_pickle=True to unpickle it fully (only for directories you trust)") from e
    return torch.load(path, weights_only=False)


def load_bnn(model: nn.Module, load_path: str, device=None, allow_pickle: bool = False, class_kwargs: dict | None = None,
             **ctor_kwargs) -> nn.Module:
    """Read a directory written by the reference's save_bnn (utils.py:87-94) the way its load_bnn does (utils.py:97-124): every
    nn.Linear named in meta.json is replaced by the class of that name from pb_llm_amd.quant, built from the stored fp16 weight
    and bias -- here the MI355X-backed class, on `device` (default: the replaced module's).  The reference calls
    `Class(weight, bias)`; classes that need more (BinaryXnorExceptOutliersLinear: outlier_fraction, which the reference's
    dead-code loader cannot supply) take it from ctor_kwargs -- each class only gets the keywords its constructor names -- or
    from class_kwargs[class name].  A directory this package's save_bnn wrote from PBLinear modules (class "PBLinear": the dense
    fp16 weight of a packed layer) is re-packed with PBLinear.from_dense, which is exact for any weight."""
    import inspect
    from . import quant as Q
    with open(os.path.join(load_path, "meta.json")) as f:
        meta = json.load(f)
    weights = _load_weights_file(os.path.join(load_path, "weights.pth"), allow_pickle)
    modules = dict(model.named_modules())
    for name, module in modules.items():
        if not isinstance(module, nn.Linear) or name not in meta:
            continue
        cls = getattr(Q, meta[name], None)
        if cls is None or not (isinstance(cls, type) and issubclass(cls, BinaryInterface)):
            raise ValueError(f"{name}: unknown binarization class {meta[name]!r}")
        w, b = weights[name + "_weight"], weights.get(name + "_bias")
        if tuple(w.shape) != (module.out_features, module.in_features):
            raise ValueError(f"{name}: stored weight is {tuple(w.shape)}, the model's Linear is {(module.out_features, module.in_features)}")
        dev = device if device is not None else module.weight.device
        b = b.data if isinstance(b, nn.Parameter) else b
        w = w.data if isinstance(w, nn.Parameter) else w
        if cls is PBLinear:
            new = PBLinear.from_dense(w.to(dev), None if b is None else b.to(dev))
        else:
            names = set(inspect.signature(cls.__init__).parameters)
            kw = {k: v for k, v in ctor_kwargs.items() if k in names}
            kw.update((class_kwargs or {}).get(meta[name], {}))
            new = cls(w.to(dev), None if b is None else b.to(dev), **kw)
        new.global_name = name.replace(".", "/")
        ind = name.rfind(".")
        father = modules[""] if ind == -1 else modules[name[:ind]]
        setattr(father, name[ind + 1:], new.to(dev))
    return model


def mask_path(low_frac, global_name: str, root: str = "gptq_pb/outputs/mask") -> str:
    """File name gptq_pb uses for a layer's low (= binarized) mask (gptq.py:111-114)."""
    return os.path.join(root, f"mask_{low_frac}_{global_name.replace('/', '_')}.pkl")


def save_low_mask(low_mask: torch.Tensor, low_frac, global_name: str, root: str = "gptq_pb/outputs/mask") -> str:
    os.makedirs(root, exist_ok=True)
    path = mask_path(low_frac, global_name, root)
    torch.save(low_mask.bool().cpu(), path)
    return path
