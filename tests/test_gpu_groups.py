"""-m gpu: column-group layers (gptq_pb --groupsize, per-(row, group) hi / lo: gptq_pb/high_quant.py:47-69 +
gptq_pb/low_quant.py) through the paths that round 1 restricted to G == 1: the matrix-core kernel (1..32 tokens, K splits
cutting through groups), the grouped and the fused decode launches (also mixed with G == 1 members), and the harness'
fuse_decode_ on a model whose linears carry groups.
"""
import numpy as np
import pytest
import torch

from oracle import pb_oracle as O
from pb_llm_amd import _lib, synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import pack_dense
from pb_llm_amd.runtime import FusedGemv, GroupedGemv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()


def T(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def assert_parity(y, ref, tol=1e-3):
    y = y.detach().float().cpu().numpy() if isinstance(y, torch.Tensor) else y
    rel, ratio = O.parity_errors(y, ref)
    assert rel < tol and ratio < 1.0, f"rel_max={rel:.3e} allclose_ratio={ratio:.3f}"


def group_layer(N, K, gs, seed, low_frac=0.9, fp16=False, exceptions=0):
    """RTN partially-binarized weight with per-(row, group) levels, packed; gs == -1: one group (per-row levels)"""
    W = synth.llm_weight(N, K, seed=seed, heavy_tail=True)
    mask = O.ptq_low_mask(W, low_frac, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    G = 1 if gs == -1 else K // gs
    hi = (r["scale"] + r["mean"]).reshape(G, N).T
    lo = (-r["scale"] + r["mean"]).reshape(G, N).T
    Wd = r["W_fq"].astype(np.float16).astype(np.float32) if fp16 else r["W_fq"].copy()
    rng = np.random.default_rng(seed)
    for _ in range(exceptions):                      # off-grid values anywhere, later groups included
        Wd[rng.integers(0, N), rng.integers(K // 2, K)] = np.float32(np.float16(rng.standard_normal()))
    p = pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=fp16)
    assert p.G == G
    return p, Wd


@pytest.mark.parametrize("N,K,gs,fp16,exc", [(100, 1536, 128, False, 3), (64, 2048, 512, True, 0), (33, 640, 128, True, 2),
                                              (48, 2048, 1024, False, 4), (4096, 4096, 128, False, 0), (1000, 4096, 256, True, 5),
                                              (16, 768, 256, False, 0)])
def test_matrix_core_kernel_with_column_groups(N, K, gs, fp16, exc):
    """every token count class of the kernel (one / two token blocks, ragged), with and without the K split (whose
    boundaries fall inside groups when gs > 256), fp32 and fp16 outputs, bias, exceptions in later groups; deterministic"""
    p, Wd = group_layer(N, K, gs, seed=N + K + gs, fp16=fp16, exceptions=exc)
    assert p.nexc >= exc // 2
    pd = p.to(DEV)
    b = synth.normal((N,), 4, 3, 0.1)
    for M in (1, 5, 16, 17, 32):
        x = synth.activations((M, K), N + M, 21)
        ref = O.dense_linear(x, Wd, b)
        for split in (True, False):
            y32 = Q.mfma_forward(pd, T(b), T(x), out_f32=True, split=split)
            assert_parity(y32, ref, 2e-4)
        assert torch.equal(y32, Q.mfma_forward(pd, T(b), T(x), out_f32=True, split=False))
        assert_parity(Q.mfma_forward(pd, None, T(x)), O.dense_linear(x, Wd))


def test_module_routes_group_layers_to_the_matrix_core_kernel():
    """PBLinear on a group layer: <= 2 tokens the column-group GEMV, 3..32 the matrix-core kernel (large layers; small ones
    from 9 tokens), above that the dense workspace -- all equal to the oracle, and the three agree with each other"""
    p, Wd = group_layer(2048, 1024, 128, seed=77, fp16=True)
    layer = Q.PBLinear(p.to(DEV), None)
    for M in (1, 2, 3, 8, 9, 32, 33, 70):
        x = synth.activations((M, 1024), 3 + M, 21)
        assert_parity(layer(T(x)), O.dense_linear(x, Wd))
    x = synth.activations((6, 1024), 9, 21)
    np.testing.assert_allclose(layer(T(x)).float().cpu().numpy(),
                               torch.cat([layer(T(x[:2])), layer(T(x[2:4])), layer(T(x[4:]))]).float().cpu().numpy(), rtol=2e-3, atol=2e-3)
    # a group size the matrix-core kernel does not take (not a power of two): the kernel refuses, the module still answers
    p3, Wd3 = group_layer(64, 1152, 384, seed=5)
    with pytest.raises(_lib.PblError):
        Q.mfma_forward(p3.to(DEV), None, T(synth.activations((8, 1152), 1, 21)))
    x3 = synth.activations((20, 1152), 2, 21)
    assert_parity(Q.PBLinear(p3.to(DEV), None)(T(x3)), O.dense_linear(x3, Wd3))


@pytest.mark.parametrize("M", [1, 2, 3, 4])
def test_fused_launch_with_column_groups(M):
    """q/k/v-style fusion where members carry groups of different sizes, one member has none, one is an fp16 checkpoint"""
    K = 1024
    specs = [(96, 128, False), (160, 256, True), (40, -1, False), (64, 1024 // 2, False)]
    ps, refs, bs = [], [], []
    x = synth.activations((M, K), 5, 21)
    for i, (N, gs, fp16) in enumerate(specs):
        p, Wd = group_layer(N, K, gs, seed=30 + i, fp16=fp16, exceptions=i)
        b = synth.normal((N,), 6 + i, 3, 0.1) if i % 2 == 0 else None
        ps.append(p); bs.append(T(b) if b is not None else None); refs.append(O.dense_linear(x, Wd, b))
    f = FusedGemv(ps, bs, DEV)
    assert f.flags & 1
    outs = f(T(x))
    assert [tuple(o.shape) for o in outs] == [(M, s[0]) for s in specs]
    for i, (o, ref) in enumerate(zip(outs, refs)):
        assert torch.isfinite(o).all(), i
        assert_parity(o, ref)
    outs32 = f(T(x), out_f32=True)
    for o, ref, p, b in zip(outs32, refs, f.packed, f.biases):
        assert_parity(o, ref, 2e-4)
        # the member's own launch runs the same kernel on the same record: bit for bit
        ys = torch.empty(M, p.N, dtype=torch.float32, device=DEV)
        import ctypes as C
        layer = p.layer_struct(b)
        if p.G > 1 and M <= 2:
            _lib.check(_lib.lib().pbl_linear_f16(C.byref(layer), T(x).data_ptr(), ys.data_ptr(), M, 1,
                                                 torch.cuda.current_stream().cuda_stream), "linear")
            assert torch.equal(ys, o)


def test_grouped_launch_with_column_groups():
    """independent layers (own x each, different K) in one launch, 4 tokens: two passes of the 2-token column-group kernel"""
    specs = [(200, 1024, 128), (64, 2048, 256), (48, 512, -1), (130, 1536, 512)]
    packed, xs, refs = [], [], []
    for i, (N, K, gs) in enumerate(specs):
        p, Wd = group_layer(N, K, gs, seed=60 + i, exceptions=1)
        packed.append(p)
        xs.append(synth.activations((4, K), 70 + i, 21))
        refs.append(O.dense_linear(xs[-1], Wd))
    grp = GroupedGemv(packed, None, M=4, device=DEV)
    assert grp.any_groups & 1
    for t, x in zip(grp.x, xs):
        t.copy_(T(x))
    for i, (y, ref) in enumerate(zip(grp.launch(), refs)):
        assert torch.isfinite(y).all(), i
        assert_parity(y, ref)


def test_fuse_decode_on_a_model_with_column_groups():
    """harness.fuse_decode_ no longer skips members with groups: token-by-token logits equal the unfused model's"""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    from pb_llm_amd import harness as H
    torch.manual_seed(1)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8,
                      num_key_value_heads=8, vocab_size=500, max_position_embeddings=128)
    model = LlamaForCausalLM(cfg).half().eval()

    def producer(name, W):
        Wn = W.float().numpy()
        mask = O.ptq_low_mask(Wn, 0.9, "magnitude", None, 128)
        r = O.ptq_rtn(Wn, mask, 8, 128)
        return dict(W_fq=torch.from_numpy(r["W_fq"]), low_mask=torch.from_numpy(mask), hscale=r["hscale"], hzero=r["hzero"], groupsize=128)

    side = H.quantize_dense_(model, producer)
    plain = H.to_pb_(model, side).to(DEV)
    assert all(m.packed.G == m.in_features // 128 for m in plain.modules() if isinstance(m, Q.PBLinear))
    fused = copy.deepcopy(plain)
    assert H.fuse_decode_(fused) == 4
    ids = torch.from_numpy((synth.uniform01(32, 3, 1) * 500).astype(np.int64)).view(1, -1).to(DEV)
    with torch.no_grad():
        for t in range(4):
            tok = ids[:, t:t + 1]
            assert torch.equal(fused(tok, use_cache=False).logits, plain(tok, use_cache=False).logits), t
        for T_ in (2, 3, 4, 20):
            a, b = fused(ids[:, :T_], use_cache=False).logits, plain(ids[:, :T_], use_cache=False).logits
            # 3-4 rows: the fused launch runs the GEMV variant twice, the members alone the matrix-core kernel (small
            # layers: the GEMV too) -- equal to rounding
            torch.testing.assert_close(a.float(), b.float(), rtol=2e-2, atol=2e-2)


def test_zero_valued_fp16_salients_with_column_groups():
    """fp16-checkpoint layers with groups and MANY salients of value exactly 0 (code == zero point): the matrix-core tile must
    hold them as -0 (an all-zero half is an empty position whose binarized level is not cancelled)"""
    N, K, gs = 48, 1024, 256
    rng = np.random.default_rng(11)
    G = K // gs
    hi = (0.2 + 0.1 * rng.random((N, G))).astype(np.float16).astype(np.float32)
    lo = (-0.15 - 0.1 * rng.random((N, G))).astype(np.float16).astype(np.float32)
    W = np.where(rng.random((N, K)) < 0.5, np.repeat(hi, gs, 1), np.repeat(lo, gs, 1)).astype(np.float32)
    ss = np.full(N, 0.01, np.float32); sz = np.full(N, 100.0, np.float32)
    sal = rng.random((N, K)) < 0.15
    q = rng.integers(0, 256, (N, K)); q[:, ::5] = 100
    vals = (ss[:, None] * (q.astype(np.float32) - sz[:, None])).astype(np.float16).astype(np.float32)
    W[sal] = vals[sal]
    p = pack_dense(W, hi, lo, ss, sz, sal.astype(np.uint8), sal_f16=True)
    assert p.G == G and p.nexc == 0 and int((W[sal] == 0).sum()) > 500
    pd = p.to(DEV)
    for M in (1, 2, 5, 11, 16, 17, 32):
        x = synth.activations((M, K), 90 + M, 21)
        ref = O.dense_linear(x, W)
        assert_parity(Q.mfma_forward(pd, None, T(x), out_f32=True), ref, 2e-4)
        assert_parity(Q.PBLinear(pd, None)(T(x)), ref)
