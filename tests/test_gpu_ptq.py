"""-m gpu: GPTQ-PB quantisation of a layer on the device (pb_llm_amd/ptq.py, pbl_gptq_block) against the oracle's
float32 restatement of the column loop and against the goldens produced by driving the reference's LowHighGPT
(tests/golden/g5_*).

The column recurrence itself is held bit-exact against the oracle on a single block (no library GEMM involved).
A whole layer goes through rocSOLVER's Cholesky and rocBLAS GEMMs, whose rounding differs from the host LAPACK the
goldens were made with; error feedback amplifies that, so -- exactly like the oracle-vs-reference test
(tests/test_oracle_golden.py::test_g5_ptq_gptq_loop) -- the full loop is pinned by loss, quantizer parameters,
mask and element agreement rather than bit-wise.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import pb_oracle as O
from pb_llm_amd import ptq, synth
from conftest import golden
from test_oracle_golden import g5_inputs, g5_name

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32 = np.float32


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()


def T(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def oracle_block(W, U, mask, hscale, hzero, maxq, mean, scale, feedback=True):
    """one block of oracle.ptq_gptq's loop (gptq.py:129-164), float32 numpy"""
    W1 = W.astype(F32).copy()
    n = W1.shape[1]
    Q1, E1, L1 = np.zeros_like(W1), np.zeros_like(W1), np.zeros_like(W1)
    for i in range(n):
        w = W1[:, i:i + 1]
        d = U[i, i]
        m = mask[:, i:i + 1]
        q = (O.high_quantize(w, hscale, hzero, maxq) * ~m).astype(F32) + (O.low_xnor_quantize(w, mean, scale) * m).astype(F32)
        Q1[:, i] = q[:, 0]
        L1[:, i] = ((w - q) ** 2 / d ** 2)[:, 0]
        err = ((w - q) / d).astype(F32)
        if feedback:
            W1[:, i:] -= err @ U[i:i + 1, i:]
        E1[:, i] = err[:, 0]
    return Q1, E1, L1.sum(1) / 2


@pytest.mark.parametrize("N,ncols,feedback", [(64, 128, True), (5, 128, True), (33, 72, True), (64, 128, False)])
def test_gptq_block_kernel_bit_exact_vs_oracle(N, ncols, feedback):
    W = synth.llm_weight(N, ncols, seed=N + ncols, heavy_tail=True)
    W[0, 3] = 0.0
    X = synth.calib_inputs(2, 96, ncols, seed=3)
    U, _ = O.hinv_cholesky_upper(O.hessian_from_inputs(X))
    mask = O.ptq_low_mask(W, 0.8, "magnitude", None, -1)
    hscale, hzero, maxq = O.high_calibrate(W, 8)
    mean, scale = O.low_xnor_calibrate((W * mask).astype(F32))
    Q1, E1, loss = oracle_block(W, U, mask, hscale, hzero, maxq, mean, scale, feedback)
    Wd = T(W)
    losses = ptq.gptq_blocks_(Wd, T(U), T(mask), T(hscale), T(hzero), float(maxq), T(mean)[None], T(scale)[None], ncols, feedback)
    assert np.array_equal(Wd.cpu().numpy(), Q1)
    np.testing.assert_allclose(losses.cpu().numpy(), loss, rtol=2e-6)


@pytest.mark.parametrize("metric", ["magnitude", "hessian"])
def test_gptq_column_loop_whole_layer_with_the_references_cholesky_factor(metric):
    """gptq_blocks_ (pbl_gptq_block + the trailing-update GEMM) over ALL 768 columns, fed the reference's own upper Cholesky
    factor, mask and quantizer state (golden G5 now stores U): the first 128-column block equals the reference BIT FOR BIT,
    the whole layer to > 99.9 % of its fp16 weights (each block's trailing update is one fp32 GEMM whose summation order
    neither torch-CPU nor rocBLAS pins), the loss to 1e-4; the oracle's loop on the same inputs agrees to the same degree."""
    W16, Xcal, _, _ = g5_inputs()
    g = golden(g5_name(metric, -1, False, 0.9))
    U = np.ascontiguousarray(g["U"].astype(np.float32))
    mask = np.unpackbits(g["mask"])[:768 * 768].astype(bool).reshape(768, 768)
    W = W16.astype(np.float32)
    hscale, hzero, maxq = O.high_calibrate(W, 8)
    mean, scale = O.low_xnor_calibrate((W * mask).astype(F32))
    Wd = T(W.copy())
    losses = ptq.gptq_blocks_(Wd, T(U), T(mask), T(hscale), T(hzero), float(maxq), T(mean)[None], T(scale)[None], 768, True)
    got, ref = Wd.cpu().numpy().astype(np.float16), g["W_fq"]
    np.testing.assert_array_equal(got[:, :128], ref[:, :128])
    assert (got != ref).mean() < 1e-3
    assert abs(float(losses.double().sum()) - float(g["loss"])) / float(g["loss"]) < 1e-4
    Wo = W.copy()
    O.gptq_blocks(Wo, U, mask, hscale, hzero, maxq, mean[None], scale[None], 768, 128)
    assert (got != Wo.astype(np.float16)).mean() < 1e-3


def run_layer(W16, Xcal, lf, metric, gs, rtn):
    layer = nn.Linear(768, 768, bias=False)
    layer.weight.data = torch.from_numpy(W16).clone()
    layer = layer.to(DEV)
    q = ptq.LowHighGPTQ(layer, salient_metric=metric, groupsize=gs, high_bit=8, disable_gptq=rtn)
    for s in range(Xcal.shape[0]):
        q.add_batch(T(Xcal[s:s + 1]), None)            # gptq_pb/run.py:155-156: one sample per call
    info = q.fasterquant(lf, blocksize=128, percdamp=0.01)
    return layer, q, info


@pytest.mark.parametrize("metric,gs,lf", [("magnitude", -1, 0.9), ("magnitude", 128, 0.9), ("hessian", -1, 0.9), ("hessian", -1, 0.95),
                                          ("hessian", 128, 0.9)])
def test_g5_rtn_branch_on_gpu(metric, gs, lf):
    W16, Xcal, x1, x32 = g5_inputs()
    g = golden(g5_name(metric, gs, True, lf))
    layer, q, _ = run_layer(W16, Xcal, lf, metric, gs, True)
    gm = np.unpackbits(g["mask"])[:768 * 768].astype(bool).reshape(768, 768)
    mism = np.count_nonzero(q.mask.cpu().numpy() != gm)
    assert mism == 0                                                    # (round 3 allowed 64 for the hessian metric: fp32 chain)
    np.testing.assert_array_equal(q.hscale.cpu().numpy().reshape(-1), g["hscale"].reshape(-1))
    np.testing.assert_array_equal(q.hzero.cpu().numpy().reshape(-1), g["hzero"].reshape(-1))
    if metric == "magnitude":
        np.testing.assert_allclose(q.mean.cpu().numpy(), g["mean"], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(q.scale.cpu().numpy(), g["scale"], rtol=3e-6)
        assert layer.weight.dtype == torch.float16
        assert np.count_nonzero(layer.weight.data.cpu().numpy() != g["W_fq"]) <= 8
    else:
        assert np.mean(layer.weight.data.cpu().numpy() == g["W_fq"]) >= 0.999


@pytest.mark.parametrize("metric,gs", [("magnitude", -1), ("hessian", 128), ("magnitude", 128)])
def test_g5_gptq_loop_on_gpu(metric, gs):
    W16, Xcal, x1, x32 = g5_inputs()
    g = golden(g5_name(metric, gs, False, 0.9))
    layer, q, info = run_layer(W16, Xcal, 0.9, metric, gs, False)
    gm = np.unpackbits(g["mask"])[:768 * 768].astype(bool).reshape(768, 768)
    # Round 4: the Cholesky chain runs in fp64 on the GPU and is rounded once (ptq.CHOL_DTYPE): its diagonal is within 2e-6
    # of the reference's stored fp32 LAPACK factor (measured 3.6e-7, gpurun_out/r44/chol.txt; the reference's own factor is
    # 2.8e-7 of max|U| away from the fp64 chain), which pins the whole layer: masks EXACT for both metrics, >= 99.9 % of the
    # fp16 weights identical (measured 99.94 - 100 %; the rest are roundings at a grid boundary after block-wise fp32 GEMM
    # updates whose summation order no library pins), loss to 1e-4 (measured <= 1e-5).  Round 3 allowed 64 mask entries,
    # 3 % of the weights and 2 % of the loss.
    assert np.count_nonzero(q.mask.cpu().numpy() != gm) == 0
    assert abs(info["error"] - float(g["loss"])) / float(g["loss"]) < 1e-4
    np.testing.assert_allclose(q.hinv_diag.cpu().numpy(), g["hinv_diag"], rtol=2e-6)
    np.testing.assert_allclose(q.scale.cpu().numpy(), g["scale"], rtol=1e-4)
    Wq = layer.weight.data.cpu().numpy()
    assert np.mean(Wq == g["W_fq"]) >= 0.999
    # the quantised layer lowers the layer-output error on the calibration inputs relative to round-to-nearest
    Xc = Xcal.reshape(-1, 768).astype(np.float64)
    ref = Xc @ W16.astype(np.float64).T
    e_gptq = np.linalg.norm(Xc @ Wq.astype(np.float64).T - ref)
    rtn = golden(g5_name(metric, gs, True, 0.9))["W_fq"].astype(np.float64)
    assert e_gptq < np.linalg.norm(Xc @ rtn.T - ref)
    # ... and packs straight into the PB format: forward == dense F.linear on the same weights
    pb = q.to_pb().to(DEV)
    np.testing.assert_array_equal(pb.weight.cpu().numpy(), Wq)
    for x in (x1, x32):
        y = pb(T(x))
        rel, ratio = O.parity_errors(y.float().cpu().numpy(), O.dense_linear(x, Wq))
        assert rel < 1e-3 and ratio < 1.0


def test_ptq_argument_errors():
    layer = nn.Linear(256, 64, bias=False).to(DEV)
    with pytest.raises(NotImplementedError):
        ptq.LowHighGPTQ(layer, salient_metric="entropy")
    with pytest.raises(ValueError):
        ptq.LowHighGPTQ(layer, groupsize=96)
    with pytest.raises(Exception):
        ptq.LowHighGPTQ(nn.Linear(8, 8))                # CPU layer: no host path


def test_quant_sequential_tiny_llama_on_gpu():
    """gptq_pb/run.py's quant_sequential counterpart on a random-init HF LLaMA (2 layers, hidden 256): every decoder Linear
    goes through the GPU GPTQ-PB pipeline with its own calibration Hessian; GPTQ lowers the quantisation damage on the
    calibration tokens relative to round-to-nearest; pack=True yields the same logits through the packed kernels."""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    from pb_llm_amd import harness as H
    from pb_llm_amd import quant as Q
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=500, max_position_embeddings=128)
    base = LlamaForCausalLM(cfg).half().eval().to(DEV)
    ids = [torch.from_numpy((synth.uniform01(64, 7, s) * 500).astype(np.int64)).view(1, -1) for s in range(4)]
    with torch.no_grad():
        ref = torch.cat([base(i.to(DEV)).logits.float() for i in ids])

    def damage(m):
        with torch.no_grad():
            out = torch.cat([m(i.to(DEV)).logits.float() for i in ids])
        return float((out - ref).norm() / ref.norm())

    m_gptq, m_rtn, m_pack = copy.deepcopy(base), copy.deepcopy(base), copy.deepcopy(base)
    e = H.quant_sequential_(m_gptq, ids, 0.9, "hessian")
    assert len(e) == 14 and all(np.isfinite(v) and v > 0 for v in e.values())
    H.quant_sequential_(m_rtn, ids, 0.9, "hessian", disable_gptq=True)
    d_gptq, d_rtn = damage(m_gptq), damage(m_rtn)
    assert 0 < d_gptq < d_rtn, (d_gptq, d_rtn)
    # every quantised weight row is two-valued off the salient set (spot check) and lm_head is untouched
    w = m_gptq.model.layers[0].mlp.down_proj.weight.data.float().cpu().numpy()
    assert torch.equal(m_gptq.lm_head.weight, base.lm_head.weight)
    assert len(np.unique(w[0])) <= 3 + int(0.1 * w.shape[1] * 3)
    H.quant_sequential_(m_pack, ids, 0.9, "hessian", pack=True)
    assert sum(isinstance(x, Q.PBLinear) for x in m_pack.modules()) == 14
    with torch.no_grad():
        a = torch.cat([m_pack(i.to(DEV)).logits.float() for i in ids]).cpu().numpy()
        b = torch.cat([m_gptq(i.to(DEV)).logits.float() for i in ids]).cpu().numpy()
    assert O.parity_errors(a, b.astype(np.float64))[0] < 5e-3


def test_cholesky_chain_on_the_gpu_against_the_references_stored_factor():
    """H -> chol -> cholesky_inverse -> chol(upper) (gptq_pb/gptq.py:74-81) in fp64 on the GPU, rounded once, against the FULL
    upper factor U the reference computed (golden G5, fp32 CPU LAPACK): within 2e-6 of max|U| everywhere (the reference's own
    factor sits 2.8e-7 from the fp64 chain), i.e. a few fp32 ulp -- the bound the whole-layer test above rests on"""
    W16, Xcal, _, _ = g5_inputs()
    g = golden(g5_name("hessian", -1, False, 0.9))
    layer = nn.Linear(768, 768, bias=False).to(DEV)
    q = ptq.LowHighGPTQ(layer, salient_metric="hessian", groupsize=-1, high_bit=8)
    for s in range(Xcal.shape[0]):
        q.add_batch(T(Xcal[s:s + 1]), None)
    H = q.H.clone()
    idx = torch.arange(768, device=DEV)
    H[idx, idx] += 0.01 * torch.mean(torch.diag(H))
    U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H.double())), upper=True).float().cpu().numpy()
    Uref = g["U"].astype(np.float32)
    assert np.abs(U - Uref).max() <= 2e-6 * np.abs(Uref).max()
    assert ptq.CHOL_DTYPE == torch.float64
