#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (/root/reference) on CPU.

Runs only in the build container (the reference does not travel to the GPU box).
Inputs come from pb_llm_amd.synth (deterministic, torch-free) so tests regenerate
them bit-exactly; only outputs of the reference are stored, as small .npz
fixtures under tests/golden/.

Shims: `import quant` calls .cuda() at class-definition time
(quant/quantizer.py:33-34) -> Tensor.cuda becomes identity; gptq.py:176,194 call
torch.cuda.synchronize / empty_cache -> no-ops.

usage: python tools/gen_goldens.py [--only G1,G4] [--out tests/golden]
"""
import argparse
import contextlib
import hashlib
import io
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pb_llm_amd import synth  # noqa: E402

REF = "/root/reference"

torch.Tensor.cuda = lambda self, *a, **k: self
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "gptq_pb"))
import quant  # noqa: E402  (reference)
from gptq import LowHighGPT  # noqa: E402  (reference)
from low_quant import LowQuantizer  # noqa: E402
from high_quant import HighQuantizer  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(out, name, **arrs):
    path = os.path.join(out, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


# --------------------------------------------------------------------------- #
def g1_g2(out):
    """BinaryLinear / XnorBinaryLinear 768x768 + bias, x [2,5,768] fp32."""
    W = synth.llm_weight(768, 768, seed=1)
    W[3, 5] = 0.0  # three-valued sign
    b = synth.normal((768,), 1, 3, 0.1)
    x = synth.normal((2, 5, 768), 1, 5, 1.0)
    m = quant.BinaryLinear(T(W), T(b))
    y1 = m(T(x)).detach().numpy()
    m0 = quant.BinaryLinear(T(W), None)
    y1nb = m0(T(x)).detach().numpy()
    m2 = quant.XnorBinaryLinear(T(W), T(b))
    y2 = m2(T(x)).detach().numpy()
    w2 = m2.quant_weight().detach().numpy()
    save(out, "g1_binary_linear", y=y1, y_nobias=y1nb)
    save(out, "g2_xnor_binary_linear", y=y2, alpha=np.abs(w2).max(1), w_sha=np.array(sha(w2)))


def g3(out):
    """weight_quant_8bit quirks: rounded zero point, uint8 wrap, constant row."""
    W = synth.normal((8, 64), 3, 0, 0.02)
    W[1] = W[1] - 0.7            # min < -0.5 -> zero point -1
    W[2] = np.abs(W[2]) + 0.6    # min > 0.5  -> zero point +1
    W[3] = 0.25                  # constant row: range 0 -> inf/nan path
    W[4] = W[4] * 40.0           # wide row, zp = round(min) far from 0
    W[5, :] = np.linspace(-0.5, 0.5, 64, dtype=np.float32)  # ties at .5
    res = {}
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
        w = T(W).to(dt)
        with np.errstate(all="ignore"):
            sim = quant.weight_quant_8bit(w.clone(), simulated=True)
            codes = quant.weight_quant_8bit(w.clone(), simulated=False)
        res["sim_" + tag] = sim.float().numpy()
        res["codes_" + tag] = codes.numpy()
    save(out, "g3_weight_quant_8bit", W=W, **res)


def g4(out):
    """BinaryXnorExceptOutliersLinear 768x768 f=0.1 (QAT layer), fp32 and fp16 weights."""
    W = synth.llm_weight(768, 768, seed=4, heavy_tail=True)
    W[7, 9] = 0.0
    b = synth.normal((768,), 4, 3, 0.1)
    x = synth.normal((3, 768), 4, 5, 1.0)
    res = {}
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
        m = quant.BinaryXnorExceptOutliersLinear(T(W).to(dt), T(b).to(dt), 0.1)
        m.eval()
        with quiet():
            m.gen_outlier_mask()
        res[f"mask_{tag}"] = np.packbits(m.outlier_mask.numpy())
        res[f"binary_scale_{tag}"] = m.binary_scale.float().numpy()
        res[f"w_hat_{tag}"] = m.weight.data.float().numpy().astype(np.float32 if tag == "f32" else np.float16)
        res[f"outlier_nbits_{tag}"] = np.array(m.outlier_nbits)
        xt = T(x).to(dt)
        with torch.no_grad():
            res[f"y_eval_{tag}"] = m(xt).float().numpy()
            lin = m.to_regular_linear()
            res[f"regular_equal_{tag}"] = np.array(bool(torch.equal(lin(xt), m(xt))))
            res[f"w_sim_sha_{tag}"] = np.array(sha(lin.weight.data.float().numpy()))
            m.train()
            res[f"y_train_{tag}"] = m(xt).float().numpy()
            res[f"binary_scale_after_train_{tag}"] = m.binary_scale.float().numpy()
            m.eval()
            res[f"y_eval2_{tag}"] = m(xt).float().numpy()
        # outlier_scale != 1
        m2 = quant.BinaryXnorExceptOutliersLinear(T(W).to(dt), None, 0.1, outlier_scale=0.5)
        m2.eval()
        with quiet(), torch.no_grad():
            res[f"y_oscale_{tag}"] = m2(xt).float().numpy()
    save(out, "g4_pb_qat_linear", **res)


def run_ptq(W16, Xcal, low_frac, metric, groupsize, disable_gptq, high_bit=8, want_hdiag=True):
    """Drive the reference PTQ objects exactly as gptq_pb/run.py:127-169 does."""
    N, K = W16.shape
    layer = nn.Linear(K, N, bias=False)
    layer.weight.data = T(W16).clone()  # fp16, like a hub checkpoint
    layer.global_name = "golden/layer"
    lq = LowQuantizer(layer.weight, method="xnor", groupsize=groupsize)
    hq = HighQuantizer(high_bit, perchannel=True, sym=False, mse=False)
    g = LowHighGPT(layer, lq, hq, salient_metric=metric, disable_gptq=disable_gptq)
    for s in range(Xcal.shape[0]):
        g.add_batch(T(Xcal[s:s + 1]), None)
    stash = {}
    orig_chol = torch.linalg.cholesky

    def chol(A, *a, **k):
        r = orig_chol(A, *a, **k)
        if k.get("upper", False):
            stash["U"] = r.clone()
        return r

    torch.linalg.cholesky = chol
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "outputs"))
        os.chdir(td)
        try:
            with quiet():
                info = g.fasterquant(low_frac, blocksize=128, percdamp=0.01)
            mask = torch.load(os.path.join(td, "outputs/mask",
                                           f"mask_{low_frac}_golden_layer.pkl")).numpy()
        finally:
            os.chdir(cwd)
            torch.linalg.cholesky = orig_chol
    r = dict(mask=mask, W_fq=layer.weight.data.numpy().copy(), mean=lq.mean.numpy().copy(),
             scale=lq.scale.numpy().copy(), hscale=hq.scale.numpy().copy(),
             hzero=hq.zero.numpy().copy(), loss=np.array(info["error"]))
    if want_hdiag:
        r["hinv_diag"] = torch.diag(stash["U"]).numpy().copy()
        r["U"] = stash["U"].numpy().copy()
    return r


def g5(out):
    """PTQ 768x768 (OPT-125m q_proj shape): {magnitude,hessian} x {gs -1,128} x {RTN,GPTQ}."""
    W16 = synth.llm_weight(768, 768, seed=5, heavy_tail=True).astype(np.float16)
    Xcal = synth.calib_inputs(4, 256, 768, seed=5)
    x1 = synth.activations((1, 768), 5, 21)
    x32 = synth.activations((32, 768), 5, 22)
    for metric in ("magnitude", "hessian"):
        for gs in (-1, 128):
            for rtn in (True, False):
                for lf in ((0.9, 0.95) if (metric == "hessian" and gs == -1 and rtn) else (0.9,)):
                    r = run_ptq(W16, Xcal, lf, metric, gs, rtn)
                    Wfq = T(r["W_fq"])
                    y1 = F.linear(T(x1), Wfq).numpy()
                    y32 = F.linear(T(x32), Wfq).numpy()
                    y32_f32 = F.linear(T(x32).float(), Wfq.float()).numpy()
                    tag = f"g5_ptq_{metric}_gs{gs if gs > 0 else 'all'}_{'rtn' if rtn else 'gptq'}_lf{lf}"
                    extra = {}
                    if not rtn and gs == -1:
                        # the full upper Cholesky factor of H^-1 (768^2 fp32) for the two GPTQ runs without groups: with it
                        # gptq.py:129-168 is reproducible column by column over the WHOLE layer, not on the diagonal only
                        extra["U"] = r["U"]
                    save(out, tag, mask=np.packbits(r["mask"]), W_fq=r["W_fq"], mean=r["mean"],
                         scale=r["scale"], hscale=r["hscale"], hzero=r["hzero"], loss=r["loss"],
                         hinv_diag=r["hinv_diag"], y1=y1, y32=y32, y32_f32=y32_f32, **extra)


def big_rtn(out, tag, N, K, low_frac, M, seed):
    """Large shapes: outputs only + hashes (regenerated by the oracle, which g5 validates)."""
    W16 = synth.llm_weight(N, K, seed=seed).astype(np.float16)
    Xcal = synth.calib_inputs(1, 8, K, seed=seed)  # H is unused by magnitude+RTN but must be PD
    r = run_ptq(W16, Xcal, low_frac, "magnitude", -1, True, want_hdiag=False)
    x = synth.activations((M, K), seed, 21)
    y = F.linear(T(x), T(r["W_fq"])).numpy()
    y_f32 = F.linear(T(x).float(), T(r["W_fq"]).float()).numpy()
    nnz_row = (~r["mask"]).sum(1).astype(np.int32)
    save(out, tag, y=y, y_f32=y_f32, nnz_row=nnz_row, W_fq_sha=np.array(sha(r["W_fq"])),
         mask_sha=np.array(sha(np.packbits(r["mask"]))), hscale=r["hscale"], hzero=r["hzero"],
         mean=r["mean"], scale=r["scale"])


def g6(out):
    big_rtn(out, "g6_llama7b_qproj_4096_lf0.9", 4096, 4096, 0.9, 1, seed=6)


def g7(out):
    big_rtn(out, "g7_llama13b_ffn_13824x5120_lf0.8", 13824, 5120, 0.8, 32, seed=7)
    big_rtn(out, "g7_llama13b_ffn_5120x13824_lf0.8", 5120, 13824, 0.8, 32, seed=8)


def g8(out):
    """One QAT training step (forward + backward) through the reference modules on CPU:
    BinaryXnorExceptOutliersLinear (train_outlier False / True, outlier_scale 1 / 0.5, fp32 and bf16-autocast),
    BinaryLinear and XnorBinaryLinear (straight-through estimator)."""
    N, K, M = 96, 320, 5
    W = synth.llm_weight(N, K, seed=8, heavy_tail=True)
    W[3, 11] = 0.0
    b = synth.normal((N,), 8, 3, 0.1)
    x = synth.normal((2, M, K), 8, 5, 1.0)
    dy = synth.normal((2, M, N), 8, 6, 1.0)
    res = dict(W=W, b=b, x=x, dy=dy)
    for tag, kw in (("base", {}), ("train_outlier", dict(train_outlier=True, outlier_scale=0.5))):
        m = quant.BinaryXnorExceptOutliersLinear(T(W), T(b), 0.1, **kw)
        m.train()
        with quiet():
            m.gen_outlier_mask()
        res[f"mask_{tag}"] = np.packbits(m.outlier_mask.numpy())
        res[f"w_hat_{tag}"] = m.weight.data.numpy().copy()
        for mode in ("f32", "bf16"):
            m.zero_grad()
            xt = T(x).clone().requires_grad_(True)
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "bf16" else contextlib.nullcontext()
            with ctx:
                y = m(xt)
            y.backward(T(dy).to(y.dtype))
            res[f"y_{tag}_{mode}"] = y.detach().float().numpy()
            res[f"dW_{tag}_{mode}"] = m.weight.grad.float().numpy().copy()
            res[f"db_{tag}_{mode}"] = m.bias.grad.float().numpy().copy()
            res[f"dx_{tag}_{mode}"] = xt.grad.float().numpy().copy()
            res[f"binary_scale_{tag}_{mode}"] = m.binary_scale.float().numpy()
    for tag, cls in (("binary", quant.BinaryLinear), ("xnor", quant.XnorBinaryLinear)):
        m = cls(T(W), T(b))
        m.train()
        xt = T(x).clone().requires_grad_(True)
        y = m(xt)
        y.backward(T(dy))
        res[f"y_{tag}"] = y.detach().numpy()
        res[f"dW_{tag}"] = m.weight.grad.numpy().copy()
        res[f"db_{tag}"] = m.bias.grad.numpy().copy()
        res[f"dx_{tag}"] = xt.grad.numpy().copy()
    save(out, "g8_qat_step", **res)


def g9(out):
    """BinaryXnorExceptOutliersLinearHessian (quant/outlier_quantizer.py:126-143) driven the way qat/run_qat.py drives it:
    the low-mask file gptq_pb dumped (gptq.py:108-114) is found under gptq_pb/outputs/mask/, becomes ~outlier_mask, the
    weights are 8-bit quantised, binary_scale stays None until a train() forward computes it.  Stored: the mask as loaded,
    W_hat, outlier_nbits, the train-mode forward and the eval forward after it, to_regular_linear's weight hash; and the
    fallback (no file -> magnitude mask) for the same weights."""
    N, K = 256, 512
    W16 = synth.llm_weight(N, K, seed=9, heavy_tail=True).astype(np.float16)
    Xcal = synth.calib_inputs(4, 128, K, seed=9)
    b = synth.normal((N,), 9, 3, 0.1)
    x = synth.normal((3, K), 9, 5, 1.0)
    res = dict(b=b)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "outputs"))
        os.makedirs(os.path.join(td, "gptq_pb", "outputs", "mask"))
        os.chdir(td)
        try:
            # 1. the reference's own PTQ run writes outputs/mask/mask_0.9_golden_layer.pkl (hessian metric, RTN values)
            layer = nn.Linear(K, N, bias=False)
            layer.weight.data = T(W16).clone()
            layer.global_name = "golden/layer"
            lq = LowQuantizer(layer.weight, method="xnor", groupsize=-1)
            hq = HighQuantizer(8, perchannel=True, sym=False, mse=False)
            g = LowHighGPT(layer, lq, hq, salient_metric="hessian", disable_gptq=True)
            for s_ in range(Xcal.shape[0]):
                g.add_batch(T(Xcal[s_:s_ + 1]), None)
            with quiet():
                g.fasterquant(0.9, blocksize=128, percdamp=0.01)
            src = os.path.join(td, "outputs", "mask", "mask_0.9_golden_layer.pkl")
            low_mask = torch.load(src).numpy()
            os.replace(src, os.path.join(td, "gptq_pb", "outputs", "mask", "mask_0.9_golden_layer.pkl"))
            res["low_mask"] = np.packbits(low_mask)
            # 2. the QAT module picks the file up (fp32 master weights, as qat/run_qat.py loads the model)
            m = quant.BinaryXnorExceptOutliersLinearHessian(T(W16).float(), T(b), 0.1)
            m.global_name = "golden/layer"
            m.eval()
            with quiet():
                m.gen_outlier_mask()
            res["outlier_mask"] = np.packbits(m.outlier_mask.numpy())
            res["binary_scale_is_none"] = np.array(m.binary_scale is None)
            res["w_hat"] = m.weight.data.numpy().copy()
            res["outlier_nbits"] = np.array(m.outlier_nbits)
            xt = T(x)
            with torch.no_grad():
                m.train()
                res["y_train"] = m(xt).numpy()
                res["binary_scale"] = m.binary_scale.numpy().copy()
                m.eval()
                res["y_eval"] = m(xt).numpy()
                res["w_sim_sha"] = np.array(sha(m.to_regular_linear().weight.data.numpy()))
            # 3. no mask file for this name: the magnitude fallback
            m2 = quant.BinaryXnorExceptOutliersLinearHessian(T(W16).float(), T(b), 0.1)
            m2.global_name = "golden/other"
            m2.eval()
            with quiet(), torch.no_grad():
                m2.gen_outlier_mask()
                res["fallback_mask"] = np.packbits(m2.outlier_mask.numpy())
                res["fallback_y_eval"] = m2(xt).numpy()
        finally:
            os.chdir(cwd)
    save(out, "g9_hessian_mask_module", **res)


def g10(out):
    """A directory written by the reference's own save_bnn (utils.py:87-94: meta.json {module name -> class name},
    weights.pth {name + "_weight": the module's weight as fp16, name + "_bias"}) for a two-layer model whose Linears were
    swapped for quant.BinaryLinear / quant.XnorBinaryLinear the way utils.py:97-124 swaps them; plus the reference modules'
    forwards after a load_bnn round trip (load_bnn rebuilds the modules from the fp16 weights)."""
    import json
    import utils as ref_utils                          # reference utils.py (save_bnn / load_bnn)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(256, 128, bias=True)
            self.blk = nn.Sequential(nn.Linear(128, 64, bias=False))

        def forward(self, x):
            return self.blk(self.fc1(x))

    W1 = synth.llm_weight(128, 256, seed=10)
    b1 = synth.normal((128,), 10, 3, 0.1)
    W2 = synth.llm_weight(64, 128, seed=11)
    W2[5, 7] = 0.0
    x = synth.normal((4, 256), 10, 5, 1.0)
    net = Net()
    net.fc1 = quant.XnorBinaryLinear(T(W1), T(b1))
    net.blk[0] = quant.BinaryLinear(T(W2), None)
    res = {}
    with tempfile.TemporaryDirectory() as td:
        with quiet():
            ref_utils.save_bnn(net, td)
        res["meta_json"] = np.array(open(os.path.join(td, "meta.json")).read())
        wts = torch.load(os.path.join(td, "weights.pth"))
        res["keys"] = np.array(json.dumps(sorted(wts)))
        for k, v in wts.items():
            res["w__" + k] = np.zeros(0, np.float16) if v is None else v.detach().numpy()
            res["none__" + k] = np.array(v is None)
        fresh = Net()
        with quiet():
            ref_utils.load_bnn(fresh, td)
        assert isinstance(fresh.fc1, quant.XnorBinaryLinear) and isinstance(fresh.blk[0], quant.BinaryLinear)
        with torch.no_grad():
            res["y_loaded"] = fresh(T(x)).numpy()
            res["y_fc1_loaded"] = fresh.fc1(T(x)).numpy()
    save(out, "g10_save_bnn_directory", **res)


ALL = dict(G1=g1_g2, G3=g3, G4=g4, G5=g5, G6=g6, G7=g7, G8=g8, G9=g9, G10=g10)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    sel = [s for s in a.only.split(",") if s] or list(ALL)
    for k in sel:
        print(k)
        ALL[k](a.out)
