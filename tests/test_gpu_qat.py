"""-m gpu: the QAT training step (pbl_qat_* kernels + pb_llm_amd/qat.py) against the oracle, against the
golden vectors produced by the reference's modules + autograd (tests/golden/g8_qat_step.npz), and bit-for-bit
against the reference's own elementwise arithmetic executed by torch on the same GPU.

Tolerances: fp32 step 2e-5 * max|ref| (fp32 library GEMM vs float64 oracle); bf16-autocast step 1e-2 (outputs
are bf16); w_sim and the straight-through gradient scaling are elementwise and must match torch EXACTLY.
"""
import numpy as np
import pytest
import torch

from oracle import pb_oracle as O
from pb_llm_amd import _lib, qat, synth
from pb_llm_amd import quant as Q
from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()


def T(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def relmax(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def g8_mask(g, tag):
    return np.unpackbits(g[f"mask_{tag}"])[:96 * 320].reshape(96, 320).astype(bool)


# ---------------------------------------------------------------- kernels, elementwise exactness
@pytest.mark.parametrize("wdt,odt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.float32, torch.float16),
                                     (torch.float16, torch.float16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("shape", [(256, 512), (33, 77), (1, 5)])
def test_wsim_and_wgrad_bit_exact_vs_torch(wdt, odt, shape):
    """pbl_qat_wsim / pbl_qat_wgrad == torch.where(mask, W*os, sign(W)*s).to(out) / g*coef, every bit
    (odd sizes take the scalar tail; sign(0) = 0)"""
    N, K = shape
    W = T(synth.llm_weight(N, K, seed=N + K, heavy_tail=True)).to(wdt)
    W.view(-1)[0] = 0
    mask = T(synth.normal((N, K), 3, 1, 1.0)) > 1.2
    s32 = qat.binary_scale(W, mask)
    ref_s = W[~mask].abs().mean(-1).view(-1, 1)                                   # quant/outlier_quantizer.py:90-93
    assert abs(float(s32) - float(ref_s.float())) <= 2e-3 * abs(float(ref_s.float())) if wdt != torch.float32 else \
        abs(float(s32) - float(ref_s)) <= 2e-6 * abs(float(ref_s))
    s_w = s32.to(wdt)                                                             # binary_scale lives in W's dtype
    for osc in (1.0, 0.5):
        got = qat.build_wsim(W, mask, s32, osc, odt)
        ref = torch.where(mask, W * osc, W.sign() * s_w.view(1, 1)).to(odt)       # :94-98 + autocast cast
        assert torch.equal(got, ref)
    g = T(synth.normal((N, K), 5, 2, 1.0)).to(wdt)
    for train_outlier in (False, True):
        got = qat.wgrad_(g.clone(), mask, s32, 0.5, train_outlier)
        coef = torch.where(mask, torch.tensor(0.5 if train_outlier else 0.0, device=DEV, dtype=wdt), s_w.view(1, 1).expand(N, K))
        assert torch.equal(got, g * coef)


def test_binary_scale_deterministic_and_empty_selection():
    W = T(synth.llm_weight(4096, 4096, seed=1))
    mask = W.abs() > 0.05
    a, b = qat.binary_scale(W, mask), qat.binary_scale(W, mask)
    assert torch.equal(a, b)
    ref = W[~mask].double().abs().mean()
    assert abs(float(a) - float(ref)) < 1e-6 * float(ref)
    assert torch.isnan(qat.binary_scale(W, torch.ones_like(mask)))               # torch: mean of nothing is nan


def test_qat_needs_gpu_and_valid_dtypes():
    W = torch.zeros(8, 8)
    with pytest.raises(_lib.PblError):
        qat.binary_scale(W, W > 0)
    m = Q.BinaryXnorExceptOutliersLinear(torch.randn(16, 32), None, 0.1)
    m.train()
    with pytest.raises(_lib.PblError):
        m(torch.randn(2, 32))
    with pytest.raises(KeyError):
        qat.binary_scale(torch.zeros(8, 8, dtype=torch.float64, device=DEV), torch.zeros(8, 8, dtype=torch.bool, device=DEV))


# ---------------------------------------------------------------- the module, against goldens and oracle
@pytest.mark.parametrize("tag,kw", [("base", {}), ("train_outlier", dict(train_outlier=True, outlier_scale=0.5))])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_g8_qat_step_module(tag, kw, mode):
    """BinaryXnorExceptOutliersLinear in train(): forward + backward == the reference module (golden) and the oracle"""
    g = golden("g8_qat_step")
    mask = g8_mask(g, tag)
    m = Q.BinaryXnorExceptOutliersLinear(T(g["W"]).cpu(), T(g["b"]).cpu(), 0.1, **kw)
    m.train()
    m.gen_outlier_mask()        # setup on the host like the golden (torch's GPU division is not bit-identical to the CPU's)
    m = m.to(DEV)
    assert np.array_equal(m.outlier_mask.cpu().numpy(), mask)
    assert np.array_equal(m.weight.data.cpu().numpy(), g[f"w_hat_{tag}"])
    x = T(g["x"]).requires_grad_(True)
    if mode == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        assert y.dtype == torch.bfloat16
    else:
        y = m(x)
    y.backward(T(g["dy"]).to(y.dtype))
    ref = O.pb_qat_step(g["x"], g["dy"], g[f"w_hat_{tag}"], mask, g["b"], gemm_bf16=mode == "bf16", **kw)
    tol = 2e-5 if mode == "f32" else 1e-2
    for name, got in (("y", y), ("dx", x.grad), ("dW", m.weight.grad), ("db", m.bias.grad)):
        assert relmax(got, ref[name]) < tol, name
        assert relmax(got, g[f"{name}_{tag}_{mode}"].astype(np.float64)) < tol, name + " vs golden"
    assert m.weight.grad.dtype == torch.float32 and x.grad.dtype == torch.float32
    assert abs(m.binary_scale.item() - float(g[f"binary_scale_{tag}_{mode}"].reshape(()))) <= 1.2e-7 * m.binary_scale.item()
    assert tuple(m.binary_scale.shape) == (1, 1)
    if not kw:
        assert not m.weight.grad[m.outlier_mask].any()
    # eval() after the step serves the refreshed scale through the packed kernel
    m.eval()
    with torch.no_grad():
        ye = m(T(g["x"]))
    assert relmax(ye, O.pb_qat_forward(g["x"], g[f"w_hat_{tag}"], mask, ref["binary_scale"], g["b"], kw.get("outlier_scale", 1.0))) < 1e-3


@pytest.mark.parametrize("tag,cls", [("binary", "BinaryLinear"), ("xnor", "XnorBinaryLinear")])
def test_g8_ste_modules_train(tag, cls):
    g = golden("g8_qat_step")
    m = getattr(Q, cls)(T(g["W"]).cpu(), T(g["b"]).cpu()).to(DEV)
    m.train()
    x = T(g["x"]).requires_grad_(True)
    y = m(x)
    y.backward(T(g["dy"]))
    for name, got in (("y", y), ("dx", x.grad), ("dW", m.weight.grad), ("db", m.bias.grad)):
        assert relmax(got, g[f"{name}_{tag}"].astype(np.float64)) < 2e-5, name


@pytest.mark.parametrize("wdt", [torch.float32, torch.float16])
def test_qat_step_llama_shape_vs_reference_arithmetic_on_gpu(wdt):
    """4096x4096, 64 tokens: the fused step == the reference's forward AS WRITTEN + torch autograd on the same GPU"""
    N = K = 4096
    W0 = T(synth.llm_weight(N, K, seed=12, heavy_tail=True)).to(wdt)
    mask = W0.abs() > W0.abs().float().flatten().kthvalue(int(0.9 * N * K))[0].to(wdt)
    x0 = T(synth.activations((4, 16, K), 5, 21)).to(wdt)
    dy = T(synth.normal((4, 16, N), 6, 7, 1.0)).to(wdt)

    W = W0.clone().requires_grad_(True)
    x = x0.clone().requires_grad_(True)
    y, s = qat.qat_linear(x, W, None, mask, 1.0, False)
    y.backward(dy)

    Wr = W0.clone().requires_grad_(True)
    xr = x0.clone().requires_grad_(True)
    s_ref = Wr[~mask].abs().mean(-1).view(-1, 1).detach()
    w_sim = torch.where(mask, (Wr * 1.0).detach(), qat.STEBinary.apply(Wr) * s_ref)
    yr = torch.nn.functional.linear(xr, w_sim, None)
    yr.backward(dy)
    tol = 1e-5 if wdt == torch.float32 else 2e-3
    assert abs(float(s) - float(s_ref.float())) <= tol * float(s_ref.float())
    for got, ref in ((y, yr), (x.grad, xr.grad), (W.grad, Wr.grad)):
        assert relmax(got, ref.detach().float().cpu().numpy().astype(np.float64)) < tol
    assert W.grad.dtype == wdt and not W.grad[mask].any()


def test_short_qat_training_run_tracks_reference_arithmetic():
    """8 SGD steps on a 2-layer MLP of PB layers (bf16 autocast off: fp32): weights after every step == the same
    steps taken with the reference's forward as written + torch autograd; the loss goes down; eval() afterwards
    serves the trained weights through the packed kernels."""
    torch.manual_seed(0)
    K, Hd, N, B = 256, 384, 128, 32
    W1 = synth.llm_weight(Hd, K, seed=31, heavy_tail=True) * 4
    W2 = synth.llm_weight(N, Hd, seed=32, heavy_tail=True) * 4
    l1 = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W1), None, 0.1)
    l2 = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W2), None, 0.1, train_outlier=True)
    for l in (l1, l2):
        l.gen_outlier_mask()
    model = torch.nn.Sequential(l1, torch.nn.ReLU(), l2).to(DEV).train()
    # the same thing with the reference's arithmetic
    ref_w = [l.weight.detach().clone().requires_grad_(True) for l in (l1, l2)]
    masks = [l.outlier_mask for l in (l1, l2)]
    train_out = [False, True]

    def ref_forward(x):
        h = x
        for i, (w, m) in enumerate(zip(ref_w, masks)):
            s = w[~m].abs().mean(-1).view(-1, 1).detach()
            sw = w * 1.0 if train_out[i] else (w * 1.0).detach()
            h = torch.nn.functional.linear(h, torch.where(m, sw, qat.STEBinary.apply(w) * s))
            if i == 0:
                h = torch.relu(h)
        return h

    x = T(synth.normal((B, K), 33, 1, 1.0))
    target = T(synth.normal((B, N), 33, 2, 1.0))
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    opt_ref = torch.optim.SGD(ref_w, lr=1e-2)
    losses = []
    for step in range(8):
        opt.zero_grad(); opt_ref.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x), target)
        loss.backward(); opt.step()
        loss_r = torch.nn.functional.mse_loss(ref_forward(x), target)
        loss_r.backward(); opt_ref.step()
        losses.append(float(loss.detach()))
        assert abs(float(loss.detach()) - float(loss_r.detach())) <= 1e-5 * abs(float(loss_r.detach())), step
        for l, w in zip((l1, l2), ref_w):
            assert relmax(l.weight, w.detach().cpu().numpy().astype(np.float64)) < 1e-5, step
    assert losses[-1] < losses[0]
    model.eval()
    with torch.no_grad():
        y = model(x)
        y_ref = ref_forward(x)
    assert relmax(y, y_ref.cpu().numpy().astype(np.float64)) < 2e-3


def test_modules_under_gradient_checkpointing():
    """the forward must work under gradient checkpointing (HF `gradient_checkpointing=True`; the reference's own
    modules recompute their weights with torch.utils.checkpoint, quant/quantizer.py:108,192).  A checkpointed call -- forward
    without a graph, forward again inside backward -- gives the gradients of a plain call: the QAT module in train() (dW, db, dx)
    and a packed PBLinear in eval (dx through the operator's autograd formula)."""
    from torch.utils.checkpoint import checkpoint
    g = golden("g8_qat_step")
    m = Q.BinaryXnorExceptOutliersLinear(T(g["W"]).cpu(), T(g["b"]).cpu(), 0.1)
    m.train(); m.gen_outlier_mask()
    m = m.to(DEV)
    dy = T(g["dy"])

    def grads(fn, params):
        x = T(g["x"]).requires_grad_(True)
        for p_ in params:
            p_.grad = None
        y = fn(x)
        y.backward(dy.to(y.dtype))
        return [y.detach().clone(), x.grad.clone()] + [p_.grad.clone() for p_ in params]
    plain = grads(lambda x: m(x), [m.weight, m.bias])
    ckpt = grads(lambda x: checkpoint(m, x, use_reentrant=False), [m.weight, m.bias])
    for a, b in zip(plain, ckpt):
        assert torch.equal(a, b)
    m.eval()
    packed = Q.PBLinear.from_dense(m.to_regular_linear().weight.detach().half().cpu(), T(g["b"]).cpu(), None, -1).to(DEV)
    plain = grads(lambda x: packed(x), [])
    ckpt = grads(lambda x: checkpoint(packed, x, use_reentrant=False), [])
    ck_re = grads(lambda x: checkpoint(packed, x, use_reentrant=True), [])
    for a, b, c in zip(plain, ckpt, ck_re):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert relmax(plain[1], (g["dy"].astype(np.float64) @ packed.weight.float().cpu().numpy().astype(np.float64))) < 1e-3
