"""-m gpu: parity of the HIP PB-linear (through the C ABI, libpbl.so) against the
oracle and against the golden vectors produced by the reference.

Tolerance (BASELINE.json north_star):  max|y - ref| <= 1e-3 * max|ref|
and allclose(rtol=1e-3, atol=1e-3*rms(ref)), ref = float64 F.linear on the same
operands; the goldens (reference torch CPU outputs) are held to the same bar.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pb_oracle as O
from pb_llm_amd import _lib, synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import pack_dense
from pb_llm_amd.runtime import GroupedGemv
from conftest import golden
from op_trace import LIBRARY_GEMM_OPS, called_ops
from test_oracle_golden import g5_inputs, g5_name

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


# ---------------------------------------------------------------- G1 / G2
def test_g1_binary_linear_gpu():
    W = synth.llm_weight(768, 768, seed=1); W[3, 5] = 0.0
    b = synth.normal((768,), 1, 3, 0.1)
    x = synth.normal((2, 5, 768), 1, 5, 1.0)
    g = golden("g1_binary_linear")
    m = Q.BinaryLinear(torch.from_numpy(W), torch.from_numpy(b)).to(DEV).eval()     # eval(): the packed kernels
    y = m(T(x))
    assert y.shape == (2, 5, 768) and y.dtype == torch.float32
    assert_parity(y, g["y"], 2e-5)                       # fp32 module: split-x path
    assert_parity(y, O.binary_linear_forward(x, W, b), 2e-5)
    m0 = Q.BinaryLinear(torch.from_numpy(W), None).to(DEV).eval()
    assert_parity(m0(T(x)), g["y_nobias"], 2e-5)
    assert_parity(m0(T(x).half()), O.binary_linear_forward(x.astype(np.float16), W, None))


def test_g2_xnor_binary_linear_gpu():
    W = synth.llm_weight(768, 768, seed=1); W[3, 5] = 0.0
    b = synth.normal((768,), 1, 3, 0.1)
    x = synth.normal((2, 5, 768), 1, 5, 1.0)
    g = golden("g2_xnor_binary_linear")
    m = Q.XnorBinaryLinear(torch.from_numpy(W), torch.from_numpy(b)).to(DEV).eval()
    assert_parity(m(T(x)), g["y"], 2e-5)


# ---------------------------------------------------------------- G4 (QAT layer)
@pytest.mark.parametrize("tag,dt", [("f32", torch.float32), ("f16", torch.float16)])
def test_g4_qat_layer_gpu(tag, dt):
    W = synth.llm_weight(768, 768, seed=4, heavy_tail=True); W[7, 9] = 0.0
    b = synth.normal((768,), 4, 3, 0.1)
    x = synth.normal((3, 768), 4, 5, 1.0)
    g = golden("g4_pb_qat_linear")
    m = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W).to(dt), torch.from_numpy(b).to(dt), 0.1)
    m.eval()
    m.gen_outlier_mask()
    np.testing.assert_array_equal(np.packbits(m.outlier_mask.numpy()), g[f"mask_{tag}"])
    assert abs(m.outlier_nbits - float(g[f"outlier_nbits_{tag}"])) < 1e-12
    np.testing.assert_array_equal(m.weight.data.float().numpy(), g[f"w_hat_{tag}"].astype(np.float32))
    m = m.to(DEV)
    xt = T(x).to(dt)
    tol = 1e-4 if tag == "f32" else 1e-3
    with torch.no_grad():
        y = m(xt)
        assert y.dtype == dt
        assert_parity(y, g[f"y_eval_{tag}"], tol)
        ref = O.pb_qat_forward(x.astype(np.float32 if tag == "f32" else np.float16), g[f"w_hat_{tag}"],
                               np.unpackbits(g[f"mask_{tag}"])[:768 * 768].astype(bool).reshape(768, 768),
                               m.binary_scale.cpu().numpy().astype(g[f"w_hat_{tag}"].dtype), b.astype(g[f"w_hat_{tag}"].dtype))
        assert_parity(y, ref, tol)
        # to_regular_linear: same math as a dense layer
        lin = m.to_regular_linear().to(DEV)
        assert_parity(lin(xt), g[f"y_eval_{tag}"], tol)
        # one train() forward refreshes binary_scale and it persists into eval()
        m.train()
        assert_parity(m(xt), g[f"y_train_{tag}"], tol)
        np.testing.assert_allclose(m.binary_scale.float().cpu().numpy(), g[f"binary_scale_after_train_{tag}"],
                                   rtol=3e-6 if tag == "f32" else 1e-3)
        m.eval()
        assert_parity(m(xt), g[f"y_eval2_{tag}"], tol)
        m2 = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W).to(dt), None, 0.1, outlier_scale=0.5).to(DEV)
        m2.eval()
        assert_parity(m2(xt), g[f"y_oscale_{tag}"], tol)


# ---------------------------------------------------------------- G5 (PTQ)
@pytest.mark.parametrize("metric,gs,rtn,lf", [("magnitude", -1, True, 0.9), ("magnitude", -1, False, 0.9),
                                               ("hessian", -1, True, 0.9), ("hessian", -1, True, 0.95),
                                               ("hessian", -1, False, 0.9),
                                               ("magnitude", 128, True, 0.9), ("magnitude", 128, False, 0.9),
                                               ("hessian", 128, True, 0.9), ("hessian", 128, False, 0.9)])
def test_g5_ptq_from_dense_checkpoint_gpu(metric, gs, rtn, lf):
    """The reference's own fake-quant fp16 weights, packed from the dense matrix +
    the mask file gptq_pb dumps; forward vs the reference's fp16 F.linear output."""
    _, _, x1, x32 = g5_inputs()
    g = golden(g5_name(metric, gs, rtn, lf))
    mask = np.unpackbits(g["mask"])[:768 * 768].astype(bool).reshape(768, 768)
    layer = Q.PBLinear.from_dense(torch.from_numpy(g["W_fq"]), None, torch.from_numpy(mask), gs,
                                  g["hscale"], g["hzero"]).to(DEV)
    np.testing.assert_array_equal(layer.weight.cpu().numpy(), g["W_fq"])   # exact repack of the checkpoint
    assert layer.packed.nexc <= 8, "fp16-rounded salients must stay 1-byte codes (PBL_FLAG_SAL_F16)"
    for x, key in ((x1, "y1"), (x32, "y32")):
        y = layer(T(x))
        assert_parity(y, O.dense_linear(x, g["W_fq"]))
        assert_parity(y, g[key])
    assert_parity(layer(T(x32).float()), g["y32_f32"], 2e-5)   # salients re-rounded to fp16 in-kernel: exact weights


@pytest.mark.parametrize("metric,gs", [("magnitude", -1), ("hessian", -1), ("magnitude", 128)])
def test_g5_flattened_fp16_checkpoint_gpu(metric, gs):
    """qat/eval_after_qat.py / --load_quantized see ONLY the dense fp16 matrix: structure is
    re-inferred (levels, code grid through the fp16 rounding); forward vs the reference output."""
    _, _, x1, x32 = g5_inputs()
    g = golden(g5_name(metric, gs, False, 0.9))
    layer = Q.PBLinear.from_dense(torch.from_numpy(g["W_fq"]), None, None, gs).to(DEV)
    np.testing.assert_array_equal(layer.weight.cpu().numpy(), g["W_fq"])
    assert layer.packed.nexc <= 0.02 * layer.packed.nnz   # off-grid leftovers only cost bytes
    assert_parity(layer(T(x32)), g["y32"])
    assert_parity(layer(T(x1)), O.dense_linear(x1, g["W_fq"]))


def test_g5_from_quantizers_matches_from_dense():
    W16, _, x1, _ = g5_inputs()
    g = golden(g5_name("magnitude", -1, True, 0.9))
    mask = np.unpackbits(g["mask"])[:768 * 768].astype(bool).reshape(768, 768)
    a = Q.PBLinear.from_quantizers(torch.from_numpy(W16), torch.from_numpy(mask), g["mean"], g["scale"],
                                   g["hscale"], g["hzero"], dtype=torch.float16)
    mism = np.count_nonzero(a.weight.cpu().numpy() != g["W_fq"])
    assert mism <= 8
    assert_parity(a.to(DEV)(T(x1)), g["y1"])


# ---------------------------------------------------------------- G6 (llama-7b q_proj, the headline shape)
@pytest.fixture(scope="module")
def llama7b_qproj():
    W = synth.llm_weight(4096, 4096, seed=6)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    return W, mask, r


def test_g6_llama7b_qproj_gemv(llama7b_qproj):
    W, mask, r = llama7b_qproj
    g = golden("g6_llama7b_qproj_4096_lf0.9")
    x = synth.activations((1, 4096), 6, 21)
    # (a) exact structure (fp32 W_fq with its quantizer state): no exceptions
    hi = r["scale"][0] + r["mean"][0]
    lo = -r["scale"][0] + r["mean"][0]
    p = pack_dense(r["W_fq"], hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8))
    assert p.nexc == 0 and p.nnz == int((~mask).sum())
    layer = Q.PBLinear(p.to(DEV), None)
    y = layer(T(x))
    assert_parity(y, O.dense_linear(x, r["W_fq"]))
    assert_parity(y, g["y"])          # the reference's fp16 forward on its fp16 weights
    assert_parity(y, g["y_f32"])
    # determinism: bitwise identical across launches
    assert torch.equal(y, layer(T(x)))
    # traffic accounting
    assert p.algorithmic_bytes(1) == 4096 * 4096 // 8 + 2 * p.nnz + 4 * 4096 + 8 * 4096 + 4 * 4097 + 4 * 4096
    assert p.nbytes < 1.13 * p.algorithmic_bytes(1)       # 7.3 % list format + alignment, 4.4 % slab index (matrix-core kernels only)


# ---------------------------------------------------------------- shapes, batches, edge cases
@pytest.mark.parametrize("N,K,M,bias", [(16, 512, 1, False), (40, 1000, 3, True), (33, 520, 5, True),
                                         (768, 3072, 2, True), (3072, 768, 9, False), (100, 8, 1, False),
                                         (17, 77, 4, True), (11008, 4096, 1, False), (4096, 11008, 2, False)])
def test_shapes_and_batches(N, K, M, bias):
    W = synth.llm_weight(N, K, seed=N + K, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    b = synth.normal((N,), 2, 3, 0.1) if bias else None
    x = synth.activations((M, K), N, 21)
    layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]), torch.from_numpy(b) if bias else None,
                                  torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    assert layer.packed.nexc == 0
    y = layer(T(x))
    assert y.shape == (M, N)
    assert_parity(y, O.dense_linear(x, r["W_fq"], b))


@pytest.mark.parametrize("N,K,gs,M,bias", [(33, 640, 128, 1, True), (64, 1024, 256, 3, False), (16, 512, 512, 2, True),
                                            (4096, 4096, 128, 1, False), (100, 1536, 128, 5, True)])
def test_column_groups(N, K, gs, M, bias):
    """per-(row, group) levels (gptq_pb --groupsize): exact structure from the oracle's RTN"""
    W = synth.llm_weight(N, K, seed=N + K + gs, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    G = K // gs
    hi = (r["scale"] + r["mean"]).reshape(G, N).T
    lo = (-r["scale"] + r["mean"]).reshape(G, N).T
    p = pack_dense(r["W_fq"], hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8))
    assert p.G == G and p.nexc <= 2
    b = synth.normal((N,), 2, 3, 0.1) if bias else None
    x = synth.activations((M, K), N, 21)
    y = Q.PBLinear(p.to(DEV), T(b) if bias else None)(T(x))
    assert_parity(y, O.dense_linear(x, r["W_fq"], b))
    W2 = r["W_fq"].copy()
    W2[min(5, N - 1), K // 2 + 3] = 0.123456   # an exception inside a later group
    p2 = pack_dense(W2, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8))
    assert p2.nexc >= 1
    assert_parity(Q.PBLinear(p2.to(DEV), None)(T(x)), O.dense_linear(x, W2))


def test_empty_batch_and_leading_dims():
    W = synth.llm_weight(32, 512, seed=9)
    s = np.sign(W).astype(np.float32)
    m = Q.BinaryLinear(torch.from_numpy(W), None).to(DEV).eval()
    assert m(torch.zeros(0, 512, device=DEV)).shape == (0, 32)
    x = synth.normal((2, 3, 4, 512), 9, 1)
    assert_parity(m(T(x)), O.dense_linear(x, s), 2e-5)


def test_exceptions_and_dense_salient_rows():
    N, K = 32, 2048
    rng = np.random.default_rng(1)
    W = np.where(rng.random((N, K)) < 0.5, 0.25, -0.125).astype(np.float32)
    ss = np.full(N, 0.01, np.float32); sz = np.full(N, 100.0, np.float32)
    W[1, :] = ss[1] * (rng.integers(0, 256, K).astype(np.float32) - 100)   # fully salient row
    W[0, [3, 900, 901, 2000]] = ss[0] * (np.array([7, 250, 0, 255], np.float32) - 100)
    for k in range(40):
        W[rng.integers(2, N), rng.integers(0, K)] = rng.standard_normal()  # off-grid -> exceptions
    hi = np.full((N, 1), 0.25, np.float32); lo = np.full((N, 1), -0.125, np.float32)
    p = pack_dense(W, hi, lo, ss, sz)
    assert p.nexc >= 30
    x = synth.activations((2, K), 3, 21)
    y = Q.PBLinear(p.to(DEV), None)(T(x))
    assert_parity(y, O.dense_linear(x, W))


def test_large_activations_and_zero_x():
    W = synth.llm_weight(64, 1024, seed=12)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]), None, torch.from_numpy(mask), -1,
                                  r["hscale"], r["hzero"]).to(DEV)
    x = synth.activations((2, 1024), 5, 21).astype(np.float32)
    x[0, ::97] *= 300.0                     # massive-activation channels
    x16 = x.astype(np.float16)
    assert_parity(layer(T(x16)), O.dense_linear(x16, r["W_fq"]))
    z = torch.zeros(1, 1024, dtype=torch.float16, device=DEV)
    assert torch.count_nonzero(layer(z)) == 0


# ---------------------------------------------------------------- properties at full size
def test_linearity_and_row_independence_full_size(llama7b_qproj):
    W, mask, r = llama7b_qproj
    p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"],
                   r["hzero"], (~mask).astype(np.uint8))
    layer = Q.PBLinear(p.to(DEV), None)
    xa = T(synth.activations((1, 4096), 1, 21)).float()
    xb = T(synth.activations((1, 4096), 2, 21)).float()
    ya, yb, yab = layer(xa), layer(xb), layer(xa + 2 * xb)
    assert_parity(yab, (ya + 2 * yb).cpu().numpy().astype(np.float64), 2e-5)
    # M=4 batch equals four M=1 calls bit-for-bit (same accumulation order per token)
    x4 = T(synth.activations((4, 4096), 3, 21))
    y4 = layer(x4)
    for m in range(4):
        assert torch.equal(y4[m:m + 1], layer(x4[m:m + 1]))


def test_llama13b_ffn_shapes_m32():
    """BASELINE config 4 shapes (low_frac 0.8, M=32) vs the reference's own output."""
    for tag, N, K, seed in (("g7_llama13b_ffn_13824x5120_lf0.8", 13824, 5120, 7),
                            ("g7_llama13b_ffn_5120x13824_lf0.8", 5120, 13824, 8)):
        g = golden(tag)
        W = synth.llm_weight(N, K, seed=seed)
        mask = O.ptq_low_mask(W, 0.8, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"],
                       r["hzero"], (~mask).astype(np.uint8))
        assert p.nexc <= 4   # sign(w - mean) == 0 exactly (value mu) is the only source of exceptions here
        x = synth.activations((32, K), seed, 21)
        y = Q.PBLinear(p.to(DEV), None)(T(x))
        assert_parity(y, g["y_f32"])
        assert_parity(y, g["y"], 2e-3)   # reference fp16-weight output (weights rounded to fp16 there)
        # the same layer as an fp16 checkpoint holds it, through the small-batch kernel over the GEMM image (round 4;
        # quant.SMALL_BATCH_IMAGE = "1": the image is built on the first small-batch call): against the reference's fp16-weight output
        # and the kernel over the packed records
        lay = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
        old = Q.SMALL_BATCH_IMAGE
        try:
            Q.SMALL_BATCH_IMAGE = "0"
            y_rec = lay(T(x))
            assert getattr(lay.packed, "_gemm_image", None) is None
            Q.SMALL_BATCH_IMAGE = "1"
            y_img = lay(T(x))
            img = lay.packed._gemm_image[1]
            assert img is not None and torch.equal(y_img, Q.small_image_forward(lay.packed, None, T(x), img))
            assert torch.equal(y_img, lay(T(x)))                                     # repeatable
            assert_parity(y_img, g["y"], 2e-3)
            assert_parity(y_img, y_rec.float().cpu().numpy().astype(np.float64), 2e-3)
            x5 = T(x)[:5].contiguous()
            assert torch.equal(lay(x5), Q.small_image_forward(lay.packed, None, x5, img))  # 5 rows: still the image kernel
            assert_parity(lay(x5), g["y"][:5], 2e-3)
            assert_parity(lay(T(x)[:4]), g["y"][:4], 2e-3)                            # 4 rows: GEMV passes over the packed records
            # 33 - 64 rows are GEMM regime for every other path; with an image they are one more pass of the small-batch kernel
            x48 = synth.activations((48, K), seed + 1, 21)
            y48 = lay(T(x48))
            assert y48.shape == (48, N) and torch.equal(y48, Q.small_image_forward(lay.packed, None, T(x48), img))
            rows = np.arange(0, N, 37)
            W16 = r["W_fq"].astype(np.float16).astype(np.float32)
            assert_parity(y48[:, torch.from_numpy(rows).to(DEV)], O.dense_linear(x48, W16[rows]))
            Q.SMALL_BATCH_IMAGE = "0"
            assert_parity(lay(T(x48)), y48.float().cpu().numpy().astype(np.float64), 2e-3)   # ... and without: the GEMM kernel over the same image
        finally:
            Q.SMALL_BATCH_IMAGE = old


# ---------------------------------------------------------------- GEMM regime (device unpack + library GEMM)
@pytest.mark.parametrize("gs,f16", [(-1, False), (-1, True), (128, False), (128, True)])
def test_device_unpack_matches_host_unpack(gs, f16):
    N, K = 100, 1536
    W = synth.llm_weight(N, K, seed=31, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    Wd = r["W_fq"].astype(np.float16).astype(np.float32) if f16 else r["W_fq"].copy()
    Wd[7, 700] = 0.3333 if not f16 else np.float32(np.float16(0.3333))          # an exception
    G = 1 if gs == -1 else K // gs
    from pb_llm_amd.packing import infer_levels
    hi, lo = infer_levels(Wd, gs, mask)
    p = pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=f16).to(DEV)
    assert p.nexc >= 1 and p.G == G
    W32 = Q.unpack_on_device(p, torch.float32).cpu().numpy()
    np.testing.assert_array_equal(W32, Wd)
    if f16:
        W16 = Q.unpack_on_device(p, torch.float16).cpu().numpy()
        np.testing.assert_array_equal(W16, Wd.astype(np.float16))


@pytest.mark.parametrize("M", [12, 64, 257])
def test_gemm_regime_forward(M, llama7b_qproj):
    W, mask, r = llama7b_qproj
    W16 = torch.from_numpy(r["W_fq"]).half()
    layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    x = synth.activations((M, 4096), 9, 21)
    y = layer(T(x))
    assert y.shape == (M, 4096) and y.dtype == torch.float16
    assert_parity(y, O.dense_linear(x, W16.numpy()))
    # the GEMV path (M <= 4) and the batched path agree on the same tokens
    y_small = torch.cat([layer(T(x[i:i + 4])) for i in range(0, 12, 4)])
    assert_parity(y[:12], y_small.float().cpu().numpy().astype(np.float64), 2e-3)


@pytest.mark.parametrize("N,K,gs,M,bias", [(4096, 4096, -1, 32, False), (100, 1536, -1, 12, True), (33, 640, 128, 17, True),
                                            (768, 3072, -1, 64, False), (5120, 13824, -1, 33, False)])
def test_small_batch_dispatch(N, K, gs, M, bias):
    """5 <= M <= 64 on fp16 checkpoints through the module: matrix-core kernel up to 32 tokens (column-group
    layers: GEMV passes, then dense), dense workspace + library GEMM above"""
    W = synth.llm_weight(N, K, seed=N + K + M, heavy_tail=True)
    lf = 0.8 if N * K > 5e7 else 0.9
    mask = O.ptq_low_mask(W, lf, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    W16 = torch.from_numpy(r["W_fq"]).half()
    W16[min(5, N - 1), K // 2 + 1] = 0.4321           # an exception
    b = synth.normal((N,), 2, 3, 0.1) if bias else None
    layer = Q.PBLinear.from_dense(W16, T(b).cpu() if bias else None, torch.from_numpy(mask), gs, r["hscale"], r["hzero"]).to(DEV)
    assert layer.packed.flags & _lib.PBL_FLAG_SAL_F16 and layer.packed.nexc >= 1
    x = synth.activations((M, K), N, 21)
    y = layer(T(x))
    assert y.shape == (M, N) and y.dtype == torch.float16
    assert_parity(y, O.dense_linear(x, W16.numpy(), b))
    # against the reference's own GPU arithmetic on the dense checkpoint (F.linear fp16)
    ref_gpu = torch.nn.functional.linear(T(x), W16.to(DEV), T(b).half() if bias else None)
    assert_parity(y, ref_gpu.float().cpu().numpy().astype(np.float64), 2e-3)


@pytest.mark.parametrize("N,K,M,kind,bias", [(4096, 4096, 32, "fp32", False), (4096, 4096, 7, "fp16", False),
                                              (100, 1536, 12, "fp16", True), (33, 520, 17, "fp32", True),
                                              (768, 3072, 1, "qat", True), (5120, 13824, 32, "fp16", False)])
def test_mfma_kernel(N, K, M, kind, bias):
    """pbl_gemm_mfma_f16: matrix-core kernel for 1..32 tokens, any G == 1 layer"""
    W = synth.llm_weight(N, K, seed=N + K + M, heavy_tail=True)
    b = synth.normal((N,), 2, 3, 0.1) if bias else None
    bt = T(b) if bias else None
    if kind == "qat":
        m = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W), None, 0.1)
        m.eval(); m.gen_outlier_mask()
        p = m._pack().to(DEV)
        Wd = m.binarize_except_outliers().numpy()
    else:
        lf = 0.8 if N * K > 5e7 else 0.9
        mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        Wd = r["W_fq"].copy()
        if kind == "fp16":
            Wd = Wd.astype(np.float16).astype(np.float32)
        Wd[min(5, N - 1), K // 2 + 1] = 0.4321 if kind == "fp32" else np.float32(np.float16(0.4321))
        from pb_llm_amd.packing import infer_levels
        hi, lo = infer_levels(Wd, -1, mask)
        p = pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=kind == "fp16").to(DEV)
        assert p.nexc >= 1
    x = synth.activations((M, K), N, 21)
    ref = O.dense_linear(x, Wd, b)
    y = Q.mfma_forward(p, bt, T(x))
    assert y.shape == (M, N) and y.dtype == torch.float16
    assert_parity(y, ref)
    y32 = Q.mfma_forward(p, bt, T(x), out_f32=True)
    assert_parity(y32, ref, 2e-4)
    assert torch.equal(y32, Q.mfma_forward(p, bt, T(x), out_f32=True))      # deterministic


# ---------------------------------------------------------------- grouped launch
def test_grouped_launch_matches_individual():
    shapes = [(4096, 4096), (4096, 4096), (1024, 4096), (768, 768), (11008, 4096)]
    packed, xs, refs = [], [], []
    for i, (N, K) in enumerate(shapes):
        W = synth.llm_weight(N, K, seed=20 + i)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        packed.append(pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0],
                                 r["hscale"], r["hzero"], (~mask).astype(np.uint8)))
        xs.append(synth.activations((2, K), 30 + i, 21))
        refs.append(O.dense_linear(xs[-1], r["W_fq"]))
    grp = GroupedGemv(packed, None, M=2, device=DEV)
    for t, x in zip(grp.x, xs):
        t.copy_(T(x))
    ys = grp.launch()
    torch.cuda.synchronize()
    for p, x, y, ref in zip(grp.packed, xs, ys, refs):
        assert_parity(y, ref)
        # single-layer launches of small layers run in split (latency) mode: same math, another
        # summation split, so equal to rounding rather than bit-for-bit
        assert_parity(Q.PBLinear(p, None)(T(x)), y.float().cpu().numpy().astype(np.float64), 2e-3)


def test_hipgraph_capture_and_side_stream(llama7b_qproj):
    """the C ABI is asynchronous on the stream it is given and graph-capturable (no allocation or
    synchronisation inside): capture one forward on a side stream, replay it with new inputs"""
    W, mask, r = llama7b_qproj
    p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"],
                   r["hzero"], (~mask).astype(np.uint8)).to(DEV)
    layer = Q.PBLinear(p, None)
    xs = [synth.activations((1, 4096), 40 + i, 21) for i in range(3)]
    x_static = T(xs[0]).clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        y_side = layer(x_static)              # eager on a non-default stream
    side.synchronize()
    assert_parity(y_side, O.dense_linear(xs[0], r["W_fq"]))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_static = layer(x_static)
    for x in xs:
        x_static.copy_(T(x))
        g.replay()
        torch.cuda.synchronize()
        assert_parity(y_static, O.dense_linear(x, r["W_fq"]))


def test_small_batch_image_kernel_under_hipgraph_capture(llama7b_qproj):
    """a batch of 16 rows through the small-batch kernel over the layer's GEMM image, captured and replayed: the image is built by
    an eager call first (a build reads two words back: never under capture), the captured call must not wait for the build's event"""
    W, mask, r = llama7b_qproj
    lay = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    W16 = r["W_fq"].astype(np.float16).astype(np.float32)
    xs = [synth.activations((16, 4096), 60 + i, 21) for i in range(3)]
    x_static = T(xs[0]).clone()
    old = Q.SMALL_BATCH_IMAGE
    try:
        Q.SMALL_BATCH_IMAGE = "1"
        y_eager = lay(x_static)
        img = lay.packed._gemm_image[1]
        assert img is not None and torch.equal(y_eager, Q.small_image_forward(lay.packed, None, x_static, img))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            y_static = lay(x_static)
        for x in xs:
            x_static.copy_(T(x))
            g.replay()
            torch.cuda.synchronize()
            assert_parity(y_static, O.dense_linear(x, W16))
            assert torch.equal(y_static, Q.small_image_forward(lay.packed, None, x_static, img))
    finally:
        Q.SMALL_BATCH_IMAGE = old


def test_build_gemm_images_for_a_model_then_small_batches_use_them():
    """harness.build_gemm_images_: every fp16 packed linear of a module tree gets its image up front; with the default policy
    ("auto") a batch of 16 rows then runs the small-batch kernel over it -- also when the call is captured first thing"""
    from pb_llm_amd import harness as H
    import torch.nn as nn
    W = synth.llm_weight(512, 1024, seed=3)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    mk = lambda: Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])   # noqa: E731
    model = nn.Sequential(mk(), nn.Sequential(mk())).to(DEV)
    assert Q.SMALL_BATCH_IMAGE == "1"                                               # the shipped default builds the image on first use ...
    old = Q.SMALL_BATCH_IMAGE
    try:
        Q.SMALL_BATCH_IMAGE = "auto"                                                # ... "auto" only uses one that exists
        x = T(synth.activations((16, 1024), 5, 21))
        y_rec = model[0](x)
        assert getattr(model[0].packed, "_gemm_image", None) is None                # no image yet: the kernel over the records ran
        n, nbytes = H.build_gemm_images_(model)
        assert n == 2 and nbytes > 0
        img = model[0].packed._gemm_image[1]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            y_cap = model[0](x)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(y_cap, Q.small_image_forward(model[0].packed, None, x, img))
        assert_parity(y_cap, y_rec.float().cpu().numpy().astype(np.float64), 2e-3)
        # the memory knob: drop the images again (nothing captured may still use them: this graph is done) ...
        del g
        assert H.drop_gemm_images_(model) == nbytes and getattr(model[0].packed, "_gemm_image", None) is None
        assert torch.equal(model[0](x), y_rec)                                      # ... and "auto" is back on the kernel over the records
    finally:
        Q.SMALL_BATCH_IMAGE = old


def test_misuse_raises():
    W = synth.llm_weight(16, 512, seed=1)
    m = Q.BinaryLinear(torch.from_numpy(W), None).to(DEV)
    for mode in (m.train, m.eval):              # same errors on the training and on the packed path
        mode()
        with pytest.raises(ValueError):
            m(torch.zeros(1, 100, device=DEV))
        with pytest.raises(_lib.PblError):
            m(torch.zeros(1, 512))


# ---------------------------------------------------------------- model level (HF LLaMA, random init)
def test_llama_model_forward_with_pb_linears():
    """Drop-in at model level: a random-init HF LlamaForCausalLM (2 layers, hidden 512), every
    decoder Linear quantized by the oracle's restatement of gptq_pb RTN (low_frac 0.9 + 8-bit
    salients), then swapped for PBLinear.  Logits and perplexity vs the SAME model with dense
    fake-quant fp16 weights (what gptq_pb/run.py evaluates).  T=48 tokens exercises the GEMM
    regime, a 5-token prompt the GEMV regime (HF calls each projection with [B, T, K])."""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    from pb_llm_amd import harness as H
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8,
                      num_key_value_heads=8, vocab_size=1000, max_position_embeddings=256)
    model = LlamaForCausalLM(cfg).half().eval()

    def producer(name, W):
        Wn = W.float().numpy()
        mask = O.ptq_low_mask(Wn, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(Wn, mask, 8, -1)
        return dict(W_fq=torch.from_numpy(r["W_fq"]), low_mask=torch.from_numpy(mask), hscale=r["hscale"], hzero=r["hzero"])

    side = H.quantize_dense_(model, producer)
    assert len(side) == 14 and "lm_head" not in side
    dense = copy.deepcopy(model).to(DEV)
    pb = H.to_pb_(model, side).to(DEV)
    n_pb = sum(isinstance(m, Q.PBLinear) for m in pb.modules())
    assert n_pb == 14 and isinstance(pb.lm_head, torch.nn.Linear)
    ids = torch.from_numpy((synth.uniform01(96, 5, 1) * 1000).astype(np.int64)).view(1, -1).to(DEV)
    with torch.no_grad():
        for T_ in (5, 48):
            ref = dense(ids[:, :T_]).logits.float().cpu().numpy()
            out = pb(ids[:, :T_]).logits.float().cpu().numpy()
            rel, _ = O.parity_errors(out, ref)
            assert rel < 5e-3, (T_, rel)     # two decoder layers of fp16 arithmetic stacked
        p_ref = H.perplexity(dense, ids, 48)
        p_pb = H.perplexity(pb, ids, 48)
    assert abs(p_pb - p_ref) / p_ref < 5e-3, (p_pb, p_ref)
    # flattened checkpoint: no side information at all
    flat = H.to_pb_(copy.deepcopy(dense).cpu(), None).to(DEV)
    with torch.no_grad():
        out = flat(ids[:, :5]).logits.float().cpu().numpy()
        ref5 = dense(ids[:, :5]).logits.float().cpu().numpy()
    assert O.parity_errors(out, ref5)[0] < 5e-3


def test_torch_compile_traces_pbllm_linear_op(llama7b_qproj):
    """inside torch.compile the module emits the registered op pbllm::linear (fake impl for tracing, HIP kernels at run
    time); fullgraph=True proves there is no graph break.  backend aot_eager: no code generator involved."""
    W, mask, r = llama7b_qproj
    lin = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = lin

        def forward(self, x):
            return torch.nn.functional.silu(self.proj(x)) + 1.0

    blk = Block().eval()
    seen = []

    def backend(gm, example_inputs):
        seen.extend(str(n.target) for n in gm.graph.nodes if n.op == "call_function")
        return gm.forward

    x = T(synth.activations((3, 4096), 5, 21))
    with torch.no_grad():
        eager = blk(x)
        out = torch.compile(blk, backend=backend, fullgraph=True)(x)
        out2 = torch.compile(blk, backend="aot_eager", fullgraph=True)(x)
    assert any("pbllm_native.linear" in t for t in seen), seen          # the NATIVE operator (Meta kernel for tracing), one node
    assert torch.equal(out, eager) and torch.equal(out2, eager)


def test_native_operator_all_regimes_and_input_gradient(llama7b_qproj):
    """torch.ops.pbllm_native.linear (csrc/pbl_torch.cpp) serves every row count and activation dtype, and equals the ctypes
    route (quant._pb_linear_forward) on the same inputs -- bit for bit where both run the same kernels; its Autograd kernel
    gives dx = dy @ W like the reference's fake-quant nn.Linear."""
    W, mask, r = llama7b_qproj
    lin = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), torch.from_numpy(synth.normal((4096,), 3, 3, 0.1)), torch.from_numpy(mask), -1,
                                r["hscale"], r["hzero"]).to(DEV)
    from pb_llm_amd import _lib
    assert _lib.native_linear() is not None
    p, b = lin.packed, lin.pbl_bias
    for M in (1, 3, 12, 32, 33, 300):
        for dt in (torch.float16, torch.bfloat16, torch.float32):
            x = T(synth.activations((M, 4096), 50 + M, 21)).to(dt)
            for f32 in (False, True):
                with torch.no_grad():
                    want = Q._pb_linear_forward(p, b, x, f32, None)
                got = Q.pb_linear_forward(p, b, x, out_f32=f32)
                assert got.dtype == want.dtype and got.shape == want.shape
                if M <= 32 or dt != torch.float16:
                    assert torch.equal(got, want), (M, dt, f32)                   # same kernels / same library calls on both routes
                else:                                                            # > 32 fp16 rows: the image kernel vs the ctypes route's choice
                    assert O.parity_errors(got.float().cpu().numpy(), want.float().cpu().numpy().astype(np.float64))[0] < 2e-3
    # input gradient through the operator's Autograd kernel
    Wd = lin.weight.float()
    for M, dt in ((2, torch.float16), (40, torch.float16), (5, torch.float32)):
        x = T(synth.activations((M, 4096), 7, 21)).to(dt).requires_grad_(True)
        y = lin(x)
        assert y.requires_grad
        dy = torch.randn_like(y)
        y.backward(dy)
        ref = (dy.float() @ Wd)
        assert O.parity_errors(x.grad.float().cpu().numpy(), ref.cpu().numpy().astype(np.float64))[0] < 2e-3


def test_bf16_activations_out_of_fp16_range():
    """bf16 activations beyond +-65504 or non-finite at 1, 2, 32, 33 (small-batch kernels) and 80 rows (the GEMM kernel's bf16
    epilogue), round 5: every token is scaled by a power of two ON THE DEVICE (pbl_act_bf16_prepare; exact for all finite inputs, no
    host sync), a token holding inf / NaN runs as its indicator row with scale +inf -- so the result has the reference's bf16
    F.linear pattern: large finite values computed in range, an inf giving +-inf by the weights' signs, a NaN poisoning its row, and
    ONLY its row.  The same launches eagerly and inside a captured hipGraph (round 4 needed a host sync, and under capture gave a
    NaN row for an inf token)."""
    N, K = 64, 1024
    W = synth.llm_weight(N, K, seed=2)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    bias = torch.from_numpy(synth.normal((N,), 3, 3, 0.1))
    for b in (None, bias):
        layer = Q.PBLinear.from_dense(W16, b, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
        bdev = None if b is None else b.float().to(DEV)
        for M in (1, 2, 32, 33, 80):
            x = torch.from_numpy(synth.activations((M, K), 4 + M, 21)).float().to(DEV)
            x[0, 7] = 3.0e5
            if M > 1:
                x[1, 100] = -2.0e30
            xb = x.bfloat16()
            ref = O.dense_linear(xb.float().cpu().numpy(), W16.float().numpy(), None if b is None else b.numpy())
            y = layer(xb)
            assert y.dtype == torch.bfloat16
            for t in range(M):       # per token: the scales differ by 25 orders of magnitude
                assert O.parity_errors(y[t:t + 1].float().cpu().numpy(), ref[t:t + 1])[0] < 1e-2, (M, t)
            # non-finite inputs: one inf in token 0, a NaN in token 1, -inf in the last token
            xi = xb.clone()
            xi[0, 3] = float("inf")
            if M > 1:
                xi[1, 5] = float("nan")
            if M > 2:
                xi[M - 1, 900] = float("-inf")
            want = torch.nn.functional.linear(xi.float(), W16.float().to(DEV), bdev).bfloat16()       # the reference's arithmetic
            y = layer(xi)
            assert torch.equal(torch.isnan(y), torch.isnan(want)) and torch.equal(torch.isposinf(y), torch.isposinf(want)), M
            assert torch.equal(torch.isneginf(y), torch.isneginf(want))
            assert torch.isinf(y[0]).all()                                   # the inf did propagate as inf (no weight of column 3 is 0)
            fin = torch.isfinite(xi.float()).all(dim=1)
            assert torch.isfinite(y[fin]).all()
            assert torch.equal(y[fin], layer(xb)[fin])                       # finite tokens do not notice their neighbours
            # the same call captured in a hipGraph and replayed with the non-finite input: identical bits
            xs = xb.clone()
            layer(xs)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                ys = layer(xs)
            xs.copy_(xi)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(torch.nan_to_num(ys.float(), nan=7.0), torch.nan_to_num(y.float(), nan=7.0)), M
    # the deviation the header of csrc/pbl_act.hip documents: several infinities in ONE token whose products disagree in sign
    xm = torch.zeros(1, K, dtype=torch.bfloat16, device=DEV)
    xm[0, 3], xm[0, 4] = float("inf"), float("inf")
    ym = layer(xm)
    assert not torch.isfinite(ym).any()


def _three_launch_bf16(layer, xb, out_f32=False):
    """pbl_act_bf16_prepare + pbl_linear_f16_ws (fp32 result, no bias) + pbl_act_finish: the form that serves every kernel family"""
    p = layer.packed
    x2 = xb.reshape(-1, p.K)
    xh, tsc = Q.act_bf16_prepare(x2)
    y32 = torch.empty(x2.shape[0], p.N, dtype=torch.float32, device=xb.device)
    lay = p.layer_struct(None)
    _lib.check(_lib.lib().pbl_linear_f16_ws(C.byref(lay), xh.data_ptr(), y32.data_ptr(), x2.shape[0], 1, None, 0,
                                            torch.cuda.current_stream().cuda_stream), "linear")
    return Q.act_finish(y32, tsc, layer.pbl_bias, torch.float32 if out_f32 else torch.bfloat16)


def test_bf16_single_launch_gemv_equals_the_three_launch_form(llama7b_qproj):
    """round 5: at decode (<= 4 rows) bf16 activations are converted INSIDE the GEMV (staging phase: per-token power-of-two scale
    into fp16's range; epilogue: scale back, bias, round to bf16) -- pbl_linear_bf16, what `module(x_bf16)` runs.  Bit for bit the
    arithmetic of prepare + fp16 kernel + finish, for the headline layer (throughput mode and, as an N = 512 slice, the waves-share-
    a-record mode), an fp16 checkpoint with bias, a ragged K, values beyond fp16's range and non-finite tokens; fp32 partials too."""
    W, mask, r = llama7b_qproj
    big = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    sl = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"][:512]).half(), torch.from_numpy(synth.normal((512,), 3, 3, 0.1)),
                               torch.from_numpy(mask[:512]), -1, r["hscale"][:512], r["hzero"][:512]).to(DEV)
    Wr = synth.llm_weight(80, 1000, seed=12)                       # K % 8 == 0 but not a panel multiple; N not a record multiple
    mr = O.ptq_low_mask(Wr, 0.9, "magnitude", None, -1)
    rr = O.ptq_rtn(Wr, mr, 8, -1)
    rag = Q.PBLinear.from_dense(torch.from_numpy(rr["W_fq"]).half(), None, torch.from_numpy(mr), -1, rr["hscale"], rr["hzero"]).to(DEV)
    for layer in (big, sl, rag):
        K = layer.in_features
        for M in (1, 2, 3, 4):
            x = torch.from_numpy(synth.activations((M, K), 30 + M, 21)).float().to(DEV)
            x[0, 5] = 7.0e4                                          # beyond fp16's range: the token is scaled
            xb = x.bfloat16()
            y = layer(xb)
            assert y.dtype == torch.bfloat16 and y.shape == (M, layer.out_features)
            assert torch.equal(y, _three_launch_bf16(layer, xb)), (K, M)
            ref = O.dense_linear(xb.float().cpu().numpy(), layer.weight.float().cpu().numpy(),
                                 None if layer.pbl_bias is None else layer.pbl_bias.cpu().numpy())
            assert O.parity_errors(y.float().cpu().numpy(), ref)[0] < 1e-2
            yf = Q.pb_linear_forward(layer.packed, layer.pbl_bias, xb, out_f32=True)
            assert yf.dtype == torch.float32 and torch.equal(yf, _three_launch_bf16(layer, xb, out_f32=True))
            assert torch.equal(yf.bfloat16(), y)
            xi = xb.clone()
            xi[M - 1, 9] = float("-inf")
            if M > 1:
                xi[0, 11] = float("nan")
            yi, wi = layer(xi), _three_launch_bf16(layer, xi)
            assert torch.equal(torch.nan_to_num(yi.float(), nan=3.0), torch.nan_to_num(wi.float(), nan=3.0)), (K, M)
            assert torch.isneginf(yi[M - 1]).any() or torch.isposinf(yi[M - 1]).any()
    # a strided view of a wider tensor, and a leading batch dimension
    xw = torch.from_numpy(synth.activations((2, 2 * 4096), 8, 21)).to(DEV).bfloat16()
    xv = xw[:, ::2]
    assert torch.equal(big(xv), big(xv.contiguous()))
    x3 = xw[:, :4096].reshape(2, 1, 4096)
    assert torch.equal(big(x3).reshape(2, 4096), big(x3.reshape(2, 4096)))


def test_bf16_small_batch_finish_folded_into_the_k_split_reduce():
    """round 5: bf16 activations at 5 - 64 rows over the GEMM image run prepare + small-batch kernel + reduce, the reduce applying
    the token scale, the bias and the bf16 cast (pbl_gemm_small_image_act) -- bit for bit what kernel (fp32, no bias) + pbl_act_finish
    gives, for a layer with bias, a ragged K, N not a multiple of 4, values beyond fp16's range and non-finite tokens; one split
    (no workspace) is refused before anything is launched."""
    Wq = synth.llm_weight(512, 4096, seed=5, heavy_tail=True)
    mq = O.ptq_low_mask(Wq, 0.9, "magnitude", None, -1)
    rq = O.ptq_rtn(Wq, mq, 8, -1)
    sl = Q.PBLinear.from_dense(torch.from_numpy(rq["W_fq"]).half(), torch.from_numpy(synth.normal((512,), 3, 3, 0.1)),
                               torch.from_numpy(mq), -1, rq["hscale"], rq["hzero"]).to(DEV)
    Wr = synth.llm_weight(78, 1000, seed=12)
    mr = O.ptq_low_mask(Wr, 0.9, "magnitude", None, -1)
    rr = O.ptq_rtn(Wr, mr, 8, -1)
    rag = Q.PBLinear.from_dense(torch.from_numpy(rr["W_fq"]).half(), torch.from_numpy(synth.normal((78,), 4, 3, 0.1)),
                                torch.from_numpy(mr), -1, rr["hscale"], rr["hzero"]).to(DEV)
    for layer in (sl, rag):
        p, K = layer.packed, layer.in_features
        for M in (5, 12, 32, 33, 64):
            x = torch.from_numpy(synth.activations((M, K), 60 + M, 21)).float().to(DEV)
            x[0, 5] = 7.0e4
            x[M - 1, 9] = float("-inf")
            x[2, 11] = float("nan")
            xb = x.bfloat16()
            xh, tsc = Q.act_bf16_prepare(xb)
            img, small_ok = Q._route_image(p, M, xb.dtype, True, xb.device)
            assert img is not None and small_ok
            want32 = Q.small_image_forward(p, None, xh, img, True)
            for out_f32 in (False, True):
                odt = torch.float32 if out_f32 else torch.bfloat16
                want = Q.act_finish(want32, tsc, layer.pbl_bias, odt)
                got = Q.small_image_act_forward(p, layer.pbl_bias, xh, tsc, img, odt)
                assert got is not None and got.dtype == odt, (K, M)             # (these layers split K: few rows)
                assert torch.equal(torch.nan_to_num(got.float(), nan=3.0), torch.nan_to_num(want.float(), nan=3.0)), (K, M, out_f32)
                y = Q.pb_linear_forward(p, layer.pbl_bias, xb, out_f32=out_f32)
                assert torch.equal(torch.nan_to_num(y.float(), nan=3.0), torch.nan_to_num(want.float(), nan=3.0)), (K, M, out_f32)
            y = layer(xb)                                                        # the registered operator's route: the same bits
            assert y.dtype == torch.bfloat16
            assert torch.equal(torch.nan_to_num(y.float(), nan=3.0), torch.nan_to_num(Q.act_finish(want32, tsc, layer.pbl_bias, torch.bfloat16).float(), nan=3.0))
            assert torch.isnan(y[2]).all() and torch.isinf(y[M - 1]).any()
            fin = torch.ones(M, dtype=torch.bool); fin[2] = False; fin[M - 1] = False
            ref = O.dense_linear(xb[fin.to(DEV)].float().cpu().numpy(), layer.weight.float().cpu().numpy(), layer.pbl_bias.cpu().numpy())
            assert O.parity_errors(y[fin.to(DEV)].float().cpu().numpy(), ref)[0] < 1e-2
    # without a workspace the layer runs as one split: refused, y untouched
    p = sl.packed
    xh, tsc = Q.act_bf16_prepare(torch.from_numpy(synth.activations((8, 4096), 1, 2)).to(DEV).bfloat16())
    img, _ = Q._route_image(p, 8, torch.bfloat16, True, xh.device)
    y = torch.full((8, 512), 7.0, dtype=torch.bfloat16, device=DEV)
    lay = p.layer_struct(sl.pbl_bias)
    rc = _lib.lib().pbl_gemm_small_image_act(C.byref(lay), xh.data_ptr(), y.data_ptr(), 8, _lib.PBL_DTYPE_BF16, tsc.data_ptr(), img.data.data_ptr(),
                                             img.data.numel(), img.geom, None, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == _lib.PBL_ERR_UNSUPPORTED and bool((y == 7.0).all())


def test_fp32_activations_split_and_join_on_the_device():
    """round 5: fp32 activations (the reference's fp32-only module classes) are split into two fp16 terms and joined again by one
    launch each (pbl_act_f32_split / pbl_act_f32_join) -- bit for bit the torch composition the route used before:
    hi = x.half(), lo = (x - hi.float()).half(); y = (y[:M] + y[M:]) + bias."""
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for M, K, wide in ((1, 4096, 0), (5, 1000, 0), (3, 1001, 0), (7, 4096, 8192), (33, 520, 1040)):
        xw = torch.from_numpy(synth.activations((M, wide or K), 70 + M, 21)).float().to(DEV) * 1.0009765625
        x = xw[:, :K] if wide else xw
        x[0, 3] = 1.0e5                                            # beyond fp16: hi = inf, lo = -inf (what torch gives)
        x[M - 1, 7] = 3.0e-9                                       # below fp16's subnormals: hi = 0, lo = 0
        xh = torch.empty(2 * M, K, dtype=torch.float16, device=DEV)
        _lib.check(L.pbl_act_f32_split(x.data_ptr(), M, K, K if M == 1 else x.stride(0), xh.data_ptr(), None, st), "split")
        hi = x.half()
        lo = (x - hi.float()).half()
        assert torch.equal(xh[:M].view(torch.int16), hi.view(torch.int16)), (M, K)
        assert torch.equal(xh[M:].view(torch.int16), lo.contiguous().view(torch.int16)), (M, K)
        # round 6: with tok_scale every row is first divided by s = 2^max(0, exponent(amax) - 14): rows below 2^15 keep
        # the unscaled bits with s = 1, the row holding 1e5 (exponent 16) is divided by 4 and stays finite
        tsc = torch.empty(M, dtype=torch.float32, device=DEV)
        xs2 = torch.empty_like(xh)
        _lib.check(L.pbl_act_f32_split(x.data_ptr(), M, K, K if M == 1 else x.stride(0), xs2.data_ptr(), tsc.data_ptr(), st), "split scaled")
        want_s = torch.ones(M, device=DEV)
        want_s[0] = 4.0
        assert torch.equal(tsc, want_s), (M, K, tsc)
        xd = x / want_s[:, None]
        hi2 = xd.half()
        lo2 = (xd - hi2.float()).half()
        assert torch.isfinite(xs2.float()).all()
        assert torch.equal(xs2[:M].view(torch.int16), hi2.view(torch.int16)) and torch.equal(xs2[M:].view(torch.int16), lo2.contiguous().view(torch.int16)), (M, K)
        if M > 1:
            assert torch.equal(xs2[1:M].view(torch.int16), xh[1:M].view(torch.int16)) and torch.equal(xs2[M + 1:].view(torch.int16), xh[M + 1:].view(torch.int16))
    for M, N, with_bias in ((1, 4096, True), (5, 78, True), (3, 1001, False), (40, 512, True)):
        yy = torch.from_numpy(synth.normal((2 * M, N), 5, M, 3.0)).float().to(DEV)
        b = torch.from_numpy(synth.normal((N,), 6, M, 0.5)).float().to(DEV) if with_bias else None
        want = yy[:M] + yy[M:]
        if b is not None:
            want = want + b
        for dt, code in ((torch.float32, _lib.PBL_DTYPE_F32), (torch.float16, _lib.PBL_DTYPE_F16), (torch.bfloat16, _lib.PBL_DTYPE_BF16)):
            out = torch.empty(M, N, dtype=dt, device=DEV)
            _lib.check(L.pbl_act_f32_join(yy.data_ptr(), None, b.data_ptr() if b is not None else None, M, N, out.data_ptr(), code, st), "join")
            assert torch.equal(out, want.to(dt)), (M, N, dt)
            sc = torch.ones(M, device=DEV)
            sc[M - 1] = 8.0
            _lib.check(L.pbl_act_f32_join(yy.data_ptr(), sc.data_ptr(), b.data_ptr() if b is not None else None, M, N, out.data_ptr(), code, st), "join scaled")
            want_sc = (yy[:M] + yy[M:]).double() * sc[:, None].double() + (b.double() if b is not None else 0.0)      # one rounding (fma)
            assert torch.equal(out, want_sc.float().to(dt)), (M, N, dt, "scaled")
    # the module route: a layer with bias at GEMV, small-batch and GEMM-regime row counts, both routes, against the float64 oracle
    Wq = synth.llm_weight(512, 1024, seed=5, heavy_tail=True)
    mq = O.ptq_low_mask(Wq, 0.9, "magnitude", None, -1)
    rq = O.ptq_rtn(Wq, mq, 8, -1)
    bq = synth.normal((512,), 3, 3, 0.1)
    layer = Q.PBLinear.from_dense(torch.from_numpy(rq["W_fq"]).half(), torch.from_numpy(bq), torch.from_numpy(mq), -1, rq["hscale"], rq["hzero"]).to(DEV)
    Wd = layer.weight.float().cpu().numpy()
    for M in (1, 3, 20, 40, 300):
        xf = T(synth.activations((M, 1024), 9, M)).float() * 1.0009765625
        ref = O.dense_linear(xf.cpu().numpy().astype(np.float64), Wd, bq)
        y = layer(xf)
        assert y.dtype == torch.float32 and y.shape == (M, 512)
        assert O.parity_errors(y.cpu().numpy(), ref)[0] < 2e-5, M
        y2 = Q._pb_linear_forward(layer.packed, layer.pbl_bias, xf, False, None)        # the ctypes route: the same launches
        assert torch.equal(y, y2), M
        xv = torch.cat([xf, xf], 1)[:, :1024]                                              # a strided view
        assert torch.equal(layer(xv), y), M


@pytest.mark.parametrize("M", [2, 40, 300])
def test_fp32_activations_beyond_fp16_range_and_non_finite(M):
    """the fp32 route's fp16 terms had no range handling -- |x| >= 65520 gave hi = inf, lo = -inf and a NaN row,
    an inf input gave NaN instead of +-inf, where the reference's F.linear(x_f32, w, b) (quant/quantizer.py:86,193) is finite / +-inf.
    Round 6: per-token power-of-two scaling (pbl_act_f32_split emits tok_scale, pbl_act_f32_join multiplies it back).  GEMV (2 rows),
    small-batch (40) and GEMM-regime (300) row counts, native operator and ctypes route, against the float64 oracle at 2e-5."""
    Wq = synth.llm_weight(512, 1024, seed=15, heavy_tail=True)
    mq = O.ptq_low_mask(Wq, 0.9, "magnitude", None, -1)
    rq = O.ptq_rtn(Wq, mq, 8, -1)
    bq = synth.normal((512,), 3, 13, 0.1)
    layer = Q.PBLinear.from_dense(torch.from_numpy(rq["W_fq"]).half(), torch.from_numpy(bq), torch.from_numpy(mq), -1, rq["hscale"], rq["hzero"]).to(DEV)
    Wd = layer.weight.float().cpu().numpy()
    xf = T(synth.activations((M, 1024), 19, M)).float() * 1.0009765625
    xf[0, 5] = 7.0e4                                               # one outlier beyond fp16 in an ordinary row
    xf[M - 1] *= 3.0e6                                             # a whole row far outside
    xf[M // 2, 9] = -1.5e30
    ref = O.dense_linear(xf.cpu().numpy().astype(np.float64), Wd, bq)
    for route in ("native", "ctypes"):
        y = layer(xf) if route == "native" else Q._pb_linear_forward(layer.packed, layer.pbl_bias, xf, False, None)
        assert y.dtype == torch.float32 and bool(torch.isfinite(y).all()), (route, M)
        assert O.parity_errors(y.cpu().numpy(), ref)[0] < 2e-5, (route, M)
    # an infinity: the row's outputs are +-inf by the sign of the weight it meets (NaN where that weight is 0), the other rows finite
    xi = xf.clone()
    xi[0, 11] = float("inf")
    want = torch.nn.functional.linear(xi.cpu(), torch.from_numpy(Wd), torch.from_numpy(bq))
    y = layer(xi)
    got0, want0 = y[0].cpu(), want[0]
    assert bool(torch.isinf(want0).any())
    assert torch.equal(torch.isnan(got0), torch.isnan(want0)) and torch.equal(got0[torch.isinf(want0)], want0[torch.isinf(want0)]), M
    assert bool(torch.isfinite(y[1:]).all()) and O.parity_errors(y[1:].cpu().numpy(), ref[1:])[0] < 2e-5
    xn = xf.clone()
    xn[M - 1, 0] = float("nan")
    yn = layer(xn)
    assert bool(torch.isnan(yn[M - 1]).all()) and bool(torch.isfinite(yn[:M - 1]).all())


def test_eval_forward_under_autocast_follows_f_linear():
    """F.linear is on autocast's lower-precision list (the reference's modules under the HF Trainer's bf16=True, qat/run_qat.py:120):
    fp32 / fp16 inputs are cast to the autocast dtype and the result has that dtype.  The packed modules do the same cast
    themselves (a custom operator is invisible to autocast): output dtype and numbers against F.linear on the dense weight under
    the same context, for a packed fp16 checkpoint layer and for a QAT module in eval() (fp32 weights)."""
    Wq = synth.llm_weight(512, 1024, seed=5, heavy_tail=True)
    mq = O.ptq_low_mask(Wq, 0.9, "magnitude", None, -1)
    rq = O.ptq_rtn(Wq, mq, 8, -1)
    bq = synth.normal((512,), 3, 3, 0.1)
    layer = Q.PBLinear.from_dense(torch.from_numpy(rq["W_fq"]).half(), torch.from_numpy(bq), torch.from_numpy(mq), -1, rq["hscale"], rq["hzero"]).to(DEV)
    qat = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(Wq), torch.from_numpy(bq), 0.1)
    qat.eval(); qat.gen_outlier_mask(); qat = qat.to(DEV)
    with torch.no_grad():
        for mod in (layer, qat):
            lin = mod.to_regular_linear().to(DEV)
            for M in (1, 6, 40, 300):
                xf = T(synth.activations((M, 1024), 11, M)).float()
                assert mod(xf).dtype == torch.float32                              # no autocast: F.linear's own dtype rule
                for dt in (torch.bfloat16, torch.float16):
                    with torch.autocast("cuda", dtype=dt):
                        y = mod(xf)
                        ref = torch.nn.functional.linear(xf, lin.weight.float(), lin.bias.float())
                        yh = mod(xf.half())                                       # an fp16 input is cast as well
                    assert y.dtype == dt and ref.dtype == dt and yh.dtype == dt, (M, dt)
                    yo = mod(xf.to(dt))                                           # the module on the cast input, outside the context
                    if mod is layer:                                              # hand-written kernels at every row count: the same launches
                        assert torch.equal(y, yo), (M, dt)
                    else:                                                         # (fp32-grid layer in the GEMM regime: the library GEMM on the unpacked
                        assert O.parity_errors(y.float().cpu().numpy(), yo.float().cpu().numpy().astype(np.float64))[0] < 2e-2     # weight is itself autocast)
                    assert O.parity_errors(y.float().cpu().numpy(), ref.float().cpu().numpy().astype(np.float64))[0] < (2e-2 if dt == torch.bfloat16 else 2e-3), (M, dt)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=False):
                    assert mod(xf).dtype == torch.float32


def test_bf16_fused_decode_of_a_bf16_model():
    """a bf16 HF LLaMA (how the checkpoints ship; qat/run_qat.py:120 trains under bf16) through fuse_decode_ + GraphedForward: the
    fused q/k/v and gate/up launches take bf16 activations directly (pbl_gemv_bf16_fused_host) -- logits of the fused model, eager and
    graph-replayed, equal the unfused PB model's bit for bit; against the dense bf16 model within bf16 forward tolerance"""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    from pb_llm_amd import harness as H
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8,
                      num_key_value_heads=8, vocab_size=1000, max_position_embeddings=256)
    model = LlamaForCausalLM(cfg).half().eval()

    def producer(name, W):
        Wn = W.float().numpy()
        mask = O.ptq_low_mask(Wn, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(Wn, mask, 8, -1)
        return dict(W_fq=torch.from_numpy(r["W_fq"]), low_mask=torch.from_numpy(mask), hscale=r["hscale"], hzero=r["hzero"])

    side = H.quantize_dense_(model, producer)
    dense = copy.deepcopy(model).to(DEV).bfloat16()
    plain = H.to_pb_(model, side).to(DEV)
    for m in plain.modules():                                # everything around the packed linears in bf16: norms, embeddings, lm_head
        if not isinstance(m, Q.PBLinear):
            for n_, p_ in list(m._parameters.items()):
                if p_ is not None:
                    p_.data = p_.data.bfloat16()
    plain.lm_head = plain.lm_head.bfloat16()
    fused = copy.deepcopy(plain)
    assert H.fuse_decode_(fused) == 4
    ids = torch.from_numpy((synth.uniform01(64, 7, 1) * 1000).astype(np.int64)).view(1, -1).to(DEV)
    with torch.no_grad():
        for t in range(4):
            tok = ids[:, t:t + 1]
            lp = plain(tok, use_cache=False).logits
            assert lp.dtype == torch.bfloat16
            lf = fused(tok, use_cache=False).logits
            assert torch.equal(lf, lp), t
            ld = dense(tok, use_cache=False).logits.float()
            assert float((lp.float() - ld).abs().max() / ld.abs().max()) < 6e-2, t        # bf16 end to end: 8 significand bits
        grp = fused.model.layers[0].self_attn.q_proj._group[0]
        assert grp.launches > 0 and grp.served >= 2 * grp.launches - 2           # the bf16 calls did take the fused launch
        for T_ in (3, 4, 5):
            assert torch.equal(fused(ids[:, :T_], use_cache=False).logits, plain(ids[:, :T_], use_cache=False).logits), T_
        g = H.GraphedForward(fused, ids[:, :1])
        for t in (9, 10, 11):
            tok = ids[:, t:t + 1]
            assert torch.equal(g.replay(tok), plain(tok, use_cache=False).logits), t


def test_concurrent_host_threads_on_their_own_streams():
    """SURVEY 8(b) threading: "reentrant, no global mutable state; safe to call from multiple host threads on different streams".
    Four host threads, each on its own stream, call ONE shared layer (whose GEMM image is built lazily by whoever gets there first)
    and a layer of their own at GEMV, small-batch and GEMM-regime row counts with fp16 / bf16 / fp32 activations; every result is
    bit for bit what a single thread computed before."""
    import threading
    Wq = synth.llm_weight(512, 1024, seed=5, heavy_tail=True)
    mq = O.ptq_low_mask(Wq, 0.9, "magnitude", None, -1)
    rq = O.ptq_rtn(Wq, mq, 8, -1)
    bq = synth.normal((512,), 3, 3, 0.1)

    def make():
        return Q.PBLinear.from_dense(torch.from_numpy(rq["W_fq"]).half(), torch.from_numpy(bq), torch.from_numpy(mq), -1, rq["hscale"], rq["hzero"]).to(DEV)
    first = make()
    cases = []
    for M in (1, 3, 12, 40, 300):
        x = T(synth.activations((M, 1024), 13, M))
        for xi in (x, x.bfloat16(), x.float() * 1.0009765625):
            cases.append((xi, first(xi)))
    torch.cuda.synchronize()
    shared = make()                                              # no image yet: the threads race to build it
    errors = []

    def work(tid):
        try:
            own = make()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), torch.no_grad():
                for rep in range(12):
                    for k, (xi, want) in enumerate(cases):
                        lay = shared if (k + rep + tid) % 2 else own
                        xi_s = xi.clone()                        # (allocated and written on this thread's stream)
                        y = lay(xi_s)
                        if not torch.equal(y, want):
                            errors.append((tid, rep, k, float((y.float() - want.float()).abs().max())))
            st.synchronize()
        except Exception as e:                                   # noqa: BLE001
            errors.append((tid, repr(e)))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    torch.cuda.synchronize()
    assert not errors, errors[:5]


@pytest.mark.parametrize("cls", ["binary", "xnor", "qat_f32"])
def test_fp32_grid_layers_in_the_gemm_regime_never_reach_a_library_gemm(cls):
    """VERDICT r5 item 8 / the residue of row a17: the reference's fp32-only module classes (quant/quantizer.py:75-86,172-193: weights
    forced to fp32) and the QAT layer with fp32 master weights (utils.py:34-36) ran `pbl_unpack_dev + at::linear` (fp32 library GEMM)
    above 32 kernel rows -- fp16 tiles cannot hold their values.  Round 6: two fp16 images (W = fp16(W) + 2^-12 residual) on the
    hand-written kernels + pbl_act_f32_join3.  64 and 2048 rows (small-batch and GEMM kernels), fp32 / fp16 / bf16 activations: no
    ATen GEMM operator runs (operator trace), 2e-5 against the float64 oracle for fp32 activations."""
    N, K = 768, 1024
    W = synth.llm_weight(N, K, seed=41, heavy_tail=True)
    W[3, 5] = 0.0
    b = synth.normal((N,), 41, 3, 0.1)
    if cls == "binary":
        m = Q.BinaryLinear(torch.from_numpy(W), torch.from_numpy(b)).to(DEV).eval()
        ref_fn = lambda x: O.binary_linear_forward(x, W, b)                       # noqa: E731
    elif cls == "xnor":
        m = Q.XnorBinaryLinear(torch.from_numpy(W), torch.from_numpy(b)).to(DEV).eval()
        ref_fn = lambda x: O.xnor_binary_linear_forward(x, W, b)                  # noqa: E731
    else:
        m = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W), torch.from_numpy(b), 0.1)
        m.eval()
        m.gen_outlier_mask()
        m = m.to(DEV)
        w_hat, mask, scale = m.weight.data.float().cpu().numpy(), m.outlier_mask.cpu().numpy(), m.binary_scale.cpu().numpy()
        ref_fn = lambda x: O.pb_qat_forward(x, w_hat, mask, scale, b)             # noqa: E731
    assert Q.F32_GRID_IMAGES and Q.GEMM_BACKEND == "auto"
    for M in (64, 2048):
        x = synth.normal((M, K), 41, 5 + M, 1.0)
        xt = T(x)
        with torch.no_grad():
            ops = called_ops(lambda: m(xt))
            assert not (ops & LIBRARY_GEMM_OPS), (cls, M, ops & LIBRARY_GEMM_OPS)
            y = m(xt)
            assert y.dtype == torch.float32 and y.shape == (M, N)
            assert_parity(y, ref_fn(x), 2e-5)
            assert torch.equal(y, m(xt))
            if cls != "qat_f32":
                continue                      # (the fp32-only classes raise on half inputs in the reference; ours follow the input dtype -- below for the QAT layer)
            for dt in (torch.float16, torch.bfloat16):
                xd = xt.to(dt)
                assert not (called_ops(lambda: m(xd)) & LIBRARY_GEMM_OPS), (cls, M, dt)
                yd = m(xd)
                assert yd.dtype == dt
                assert O.parity_errors(yd.float().cpu().numpy(), ref_fn(xd.float().cpu().numpy()))[0] < (1e-3 if dt == torch.float16 else 1e-2)
    # a value beyond fp16's range in the activations on this route as well
    xo = T(synth.normal((64, K), 41, 99, 1.0))
    xo[5, 7] = 2.0e5
    with torch.no_grad():
        yo = m(xo)
    assert bool(torch.isfinite(yo).all()) and O.parity_errors(yo.cpu().numpy(), ref_fn(xo.cpu().numpy()))[0] < 2e-5
    old = Q.F32_GRID_IMAGES
    try:
        Q.F32_GRID_IMAGES = False              # round 5's route: the fp32 library GEMM on the unpacked weight
        with torch.no_grad():
            assert called_ops(lambda: m(xt)) & LIBRARY_GEMM_OPS
            assert_parity(m(xt), y.cpu().numpy().astype(np.float64), 2e-5)
    finally:
        Q.F32_GRID_IMAGES = old
