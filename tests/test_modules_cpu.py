"""CPU tests of the host-side module logic that needs no kernel: mask generation against
the goldens, the Hessian-mask loader, (de)serialisation, replace_linear_with_pb."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import pb_oracle as O
from pb_llm_amd import io as pbio
from pb_llm_amd import quant as Q
from pb_llm_amd import synth
from conftest import golden


def _w():
    W = synth.llm_weight(768, 768, seed=4, heavy_tail=True)
    W[7, 9] = 0.0
    return W


def test_gen_outlier_mask_matches_reference_on_cpu():
    g = golden("g4_pb_qat_linear")
    m = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(_w()), None, 0.1)
    m.gen_outlier_mask()
    np.testing.assert_array_equal(np.packbits(m.outlier_mask.numpy()), g["mask_f32"])
    np.testing.assert_array_equal(m.weight.data.numpy(), g["w_hat_f32"])
    np.testing.assert_allclose(m.binary_scale.numpy(), g["binary_scale_f32"], rtol=3e-6)
    assert m.binary_scale.shape == (1, 1)
    assert abs(m.outlier_nbits - float(g["outlier_nbits_f32"])) < 1e-12
    # dense view used by to_regular_linear equals the oracle's w_sim
    w_sim = m.binarize_except_outliers().numpy()
    mask = np.unpackbits(g["mask_f32"])[:768 * 768].astype(bool).reshape(768, 768)
    np.testing.assert_array_equal(w_sim, O.binarize_except_outliers(g["w_hat_f32"], mask, m.binary_scale.numpy()))
    # packing reproduces it exactly (codes of the 8-bit grid, +-alpha levels, zeros as code 0)
    p = m._pack()
    np.testing.assert_array_equal(p.unpack().numpy(), w_sim)
    assert p.nexc <= 0.01 * p.nnz
    lin = m.to_regular_linear()
    assert isinstance(lin, nn.Linear) and torch.equal(lin.weight.data, torch.from_numpy(w_sim))


def test_weight_quant_8bit_host_matches_golden():
    g = golden("g3_weight_quant_8bit")
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
        w = torch.from_numpy(g["W"]).to(dt)
        ok = np.ones(8, bool); ok[3] = False
        np.testing.assert_array_equal(Q.weight_quant_8bit(w, simulated=False).numpy()[ok], g["codes_" + tag][ok])
        np.testing.assert_array_equal(Q.weight_quant_8bit(w, simulated=True).float().numpy()[ok], g["sim_" + tag][ok])


def test_hessian_variant_loads_gptq_mask(tmp_path, monkeypatch):
    """quant/outlier_quantizer.py:126-143: mask file present -> outlier_mask = ~mask and
    binary_scale stays None until a train() forward; missing -> magnitude fallback."""
    monkeypatch.chdir(tmp_path)
    W = torch.from_numpy(_w())
    low = torch.from_numpy(O.ptq_low_mask(_w(), 0.9, "magnitude"))
    m = Q.BinaryXnorExceptOutliersLinearHessian(W.clone(), None, 0.1)
    m.global_name = "model/layers/0/q_proj"
    m.gen_outlier_mask()                       # no file -> magnitude fallback
    assert m.binary_scale is not None
    pbio.save_low_mask(low, 0.9, m.global_name)
    assert os.path.exists("gptq_pb/outputs/mask/mask_0.9_model_layers_0_q_proj.pkl")
    m2 = Q.BinaryXnorExceptOutliersLinearHessian(W.clone(), None, 0.1)
    m2.global_name = m.global_name
    m2.gen_outlier_mask()
    assert torch.equal(m2.outlier_mask, ~low) and m2.binary_scale is None
    m2.train()
    m2._refresh_scale()                        # what the first train() forward does
    assert m2.binary_scale is not None and m2.binary_scale.shape == (1, 1)


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.q_proj = nn.Linear(512, 64, bias=False)
        self.mlp = nn.Sequential(nn.Linear(512, 96), nn.ReLU(), nn.Linear(96, 32))
        self.lm_head = nn.Linear(32, 10)


def _pb_factory(lin: nn.Linear):
    W = lin.weight.data.float().numpy()
    if W.shape[1] < 128:
        return Q.BinaryLinear(lin.weight.data, lin.bias.data if lin.bias is not None else None)
    mask = O.ptq_low_mask(W, 0.9, "magnitude")
    r = O.ptq_rtn(W, mask, 8, -1)
    return Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]), lin.bias, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])


def test_replace_save_load_roundtrip(tmp_path):
    torch.manual_seed(0)
    model = Tiny()
    Q.replace_linear_with_pb(model, _pb_factory, skip=("lm_head",))      # the GPTQ-PB pipeline's call: gptq_pb/modelutils.find_layers never sees lm_head
    assert isinstance(model.q_proj, Q.PBLinear) and isinstance(model.mlp[0], Q.PBLinear)
    assert isinstance(model.mlp[2], Q.BinaryLinear) and isinstance(model.lm_head, nn.Linear)
    # the default is the QAT pipeline's: qat/run_qat.py:45-66 swaps EVERY nn.Linear, lm_head included
    m2 = Tiny()
    Q.replace_linear_with_pb(m2, lambda lin: Q.BinaryLinear(lin.weight.data, lin.bias.data if lin.bias is not None else None))
    assert not any(type(m) is nn.Linear for m in m2.modules()) and isinstance(m2.lm_head, Q.BinaryLinear) and m2.lm_head.global_name == "lm_head"
    assert model.mlp[0].global_name == "mlp.0"
    model.mlp[2]._packed_on(torch.device("cpu"))
    meta = pbio.save_pb(model, str(tmp_path / "ckpt"))
    assert set(meta) == {"q_proj", "mlp.0", "mlp.2"} and meta["mlp.2"]["class"] == "BinaryLinear"
    fresh = pbio.load_pb(Tiny(), str(tmp_path / "ckpt"))
    for name in meta:
        a = dict(model.named_modules())[name]
        b = dict(fresh.named_modules())[name]
        pa = a.packed if isinstance(a, Q.PBLinear) else a._packed
        assert isinstance(b, Q.PBLinear) and torch.equal(pa.blob, b.packed.blob)
        assert torch.equal(pa.unpack(), b.packed.unpack())
        if a.bias is not None:
            assert torch.equal(a.bias.detach().float(), b.bias)
    sd = fresh.state_dict()
    assert "q_proj.pbl_blob" in sd and sd["q_proj.pbl_blob"].dtype == torch.uint8
    with pytest.raises(KeyError):
        pbio.load_pb(nn.Sequential(nn.Linear(4, 4)), str(tmp_path / "ckpt"))


def test_pbllm_linear_op_fake_impl_gives_shapes_without_a_gpu():
    """pbllm::linear has a fake (meta) implementation: tracing / torch.compile can infer shapes and dtypes with no GPU;
    the REAL implementation still refuses host tensors (no CPU compute path)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from pb_llm_amd import _lib
    W = synth.llm_weight(32, 512, seed=3)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])
    m = layer._meta
    meta = [m.N, m.K, m.P, m.G, m.NRB, m.flags, m.max_nch, m.max_nexc, m.nnz, m.nexc]
    with FakeTensorMode() as mode:
        blob = mode.from_tensor(layer.pbl_blob)
        x = mode.from_tensor(torch.zeros(2, 7, m.K, dtype=torch.float16))
        y = torch.ops.pbllm.linear(blob, None, x, meta, True, False)
        assert tuple(y.shape) == (2, 7, m.N) and y.dtype == torch.float16
        assert torch.ops.pbllm.linear(blob, None, x, meta, True, True).dtype == torch.float32
    with pytest.raises(_lib.PblError):
        torch.ops.pbllm.linear(layer.pbl_blob, None, torch.zeros(1, m.K, dtype=torch.float16), meta, True, False)


def test_training_and_producer_paths_refuse_host_tensors():
    """QAT step, salient selection, row quantizer and GPTQ-PB are GPU-only: a host tensor raises instead of silently
    running somewhere else (the same rule as the forward)."""
    from pb_llm_amd import _lib, prep, ptq, qat
    W = torch.from_numpy(synth.llm_weight(16, 256, seed=2))
    mask = W.abs() > 0.05
    for call in (lambda: qat.binary_scale(W, mask),
                 lambda: qat.build_wsim(W, mask, torch.zeros(1), 1.0),
                 lambda: qat.wgrad_(W.clone(), mask, torch.zeros(1), 1.0, False),
                 lambda: qat.qat_linear(torch.zeros(2, 256), W, None, mask),
                 lambda: prep.kth_pair(W, 1, 2),
                 lambda: prep.outlier_mask(W, torch.zeros(2)),
                 lambda: prep.quant8_rows_(W.clone()),
                 lambda: ptq.LowHighGPTQ(nn.Linear(256, 16))):
        with pytest.raises(_lib.PblError):
            call()
    m = Q.BinaryXnorExceptOutliersLinear(W.clone(), None, 0.1)
    m.train()
    with pytest.raises(_lib.PblError):
        m(torch.zeros(2, 256, requires_grad=True))
    with pytest.raises(ValueError):
        m(torch.zeros(2, 100))


def test_decoder_layers_finds_llama_and_opt_layouts():
    from transformers import LlamaConfig, LlamaForCausalLM, OPTConfig, OPTForCausalLM
    from pb_llm_amd import harness as H
    llama = LlamaForCausalLM(LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                         num_key_value_heads=4, vocab_size=100, max_position_embeddings=64))
    opt = OPTForCausalLM(OPTConfig(hidden_size=64, ffn_dim=128, num_hidden_layers=3, num_attention_heads=4, vocab_size=100,
                                   max_position_embeddings=64, word_embed_proj_dim=64))
    assert len(H.decoder_layers(llama)) == 2 and len(H.decoder_layers(opt)) == 3
    assert len(H.find_layers(H.decoder_layers(llama)[0])) == 7 and len(H.find_layers(H.decoder_layers(opt)[0])) == 6
    with pytest.raises(RuntimeError):
        H.quant_sequential_(llama, [torch.zeros(1, 8, dtype=torch.long)], 0.9)       # model on the host: no host path
    with pytest.raises(NotImplementedError):
        H.decoder_layers(nn.Linear(4, 4))
