"""-m gpu: randomized shapes through every forward path (GEMV, latency/split mode, column groups, matrix-core kernel,
dense workspace) against the float64 oracle.  Deterministic seeds (synth generators), so a failure reproduces."""
import numpy as np
import pytest
import torch

from oracle import pb_oracle as O
from oracle import pb_format_ref as FR
from pb_llm_amd import _lib, synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import pack_dense

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def case(seed):
    u = synth.uniform01(8, 1000 + seed, 1)
    N = int(1 + u[0] * 200) if seed % 3 else int(16 * (1 + int(u[0] * 40)))
    K = 8 * int(1 + u[1] * 300)                                   # 8 .. 2408, often not a multiple of 128 / 256 / 512
    sal_frac = [0.0, 0.02, 0.1, 0.3, 0.6][int(u[2] * 5) % 5]
    fp16 = u[3] < 0.5
    bias = u[4] < 0.5
    n_exc = int(u[5] * 4)
    return N, K, sal_frac, fp16, bias, n_exc


@pytest.mark.parametrize("seed", range(36))
def test_random_layer_all_token_counts(seed):
    N, K, sal_frac, fp16, bias, n_exc = case(seed)
    W = synth.llm_weight(N, K, seed=seed, heavy_tail=True)
    if sal_frac > 0:
        mask = O.ptq_low_mask(W, 1.0 - sal_frac, "magnitude", None, -1)
    else:
        mask = np.ones((N, K), bool)
    r = O.ptq_rtn(W, mask, 8, -1)
    Wd = r["W_fq"].copy()
    if fp16:
        Wd = Wd.astype(np.float16).astype(np.float32)
    for e in range(n_exc):                                         # values on neither level nor the code grid
        Wd[(7 * e) % N, (131 * e + 5) % K] = np.float32(np.float16(0.123 + e)) if fp16 else np.float32(0.123 + e)
    from pb_llm_amd.packing import infer_levels
    hi, lo = infer_levels(Wd, -1, mask)
    p = pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=fp16)
    assert np.array_equal(FR.decode(p.blob.numpy()), Wd)          # independent decoder: the blob IS the matrix
    b = synth.normal((N,), seed, 3, 0.1) if bias else None
    layer = Q.PBLinear(p.to(DEV), T(b) if bias else None, torch.float16 if fp16 else torch.float32)
    for M in (1, 3, 4, 5, 9, 17, 32, 33):
        x = synth.activations((M, K), seed, M)
        ref = O.dense_linear(x, Wd, b)
        y = layer(T(x))
        rel, ratio = O.parity_errors(y.float().cpu().numpy(), ref)
        assert rel < 1e-3 and ratio < 1.0, (seed, N, K, sal_frac, fp16, M, rel, ratio)
    # the matrix-core kernel directly at every M it accepts for this layer
    if K % 8 == 0:
        for M in (1, 2, 16, 31):
            x = synth.activations((M, K), seed, 40 + M)
            y = Q.mfma_forward(layer.packed, layer.pbl_bias, T(x), out_f32=True)
            rel, ratio = O.parity_errors(y.cpu().numpy(), O.dense_linear(x, Wd, b))
            assert rel < 2e-4 and ratio < 1.0, (seed, N, K, sal_frac, fp16, M, rel, ratio)


def _layer(seed):
    N, K, sal_frac, fp16, bias, n_exc = case(seed)
    W = synth.llm_weight(N, K, seed=seed, heavy_tail=True)
    mask = O.ptq_low_mask(W, 1.0 - sal_frac, "magnitude", None, -1) if sal_frac > 0 else np.ones((N, K), bool)
    r = O.ptq_rtn(W, mask, 8, -1)
    Wd = r["W_fq"].copy()
    if fp16:
        Wd = Wd.astype(np.float16).astype(np.float32)
    for e in range(n_exc):
        Wd[(7 * e) % N, (131 * e + 5) % K] = np.float32(np.float16(0.123 + e)) if fp16 else np.float32(0.123 + e)
    from pb_llm_amd.packing import infer_levels
    hi, lo = infer_levels(Wd, -1, mask)
    p = pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=fp16)
    b = synth.normal((N,), seed, 3, 0.1) if bias else None
    return Q.PBLinear(p.to(DEV), T(b) if bias else None, torch.float16 if fp16 else torch.float32), Wd, b, fp16


@pytest.mark.parametrize("seed", range(0, 36, 2))
def test_random_layer_gemm_regime_and_activation_dtypes(seed):
    """round 5: the same random layers beyond 33 rows (the GEMM kernel over the image with whatever launch plan the cost model picks
    -- or, for fp32-grid layers, unpack + library) and with bf16 / fp32 activations at every regime (bf16: conversion inside the GEMV
    up to 4 rows, prepare + kernel + finish up to 64, prepare + GEMM epilogue beyond; fp32: two fp16 terms) against the float64 oracle"""
    layer, Wd, b, fp16 = _layer(seed)
    N, K = Wd.shape
    for M in (70, 300):
        x = synth.activations((M, K), seed, M)
        y = layer(T(x))
        rel, ratio = O.parity_errors(y.float().cpu().numpy(), O.dense_linear(x, Wd, b))
        assert rel < 1e-3 and ratio < 1.0, (seed, N, K, fp16, M, rel, ratio)
        assert torch.equal(y, layer(T(x)))
    for M in (1, 4, 9, 33, 70, 300):
        xb = T(synth.activations((M, K), seed, 50 + M)).bfloat16()
        y = layer(xb)
        assert y.dtype == torch.bfloat16
        rel, _ = O.parity_errors(y.float().cpu().numpy(), O.dense_linear(xb.float().cpu().numpy(), Wd, b))
        assert rel < 1e-2, (seed, N, K, fp16, M, rel)                       # bf16 result: 8 significand bits
    for M in (2, 40):
        xf = T(synth.activations((M, K), seed, 90 + M)).float() * 1.0009765625     # (not representable in fp16: the low term matters)
        y = layer(xf)
        assert y.dtype == torch.float32
        rel, _ = O.parity_errors(y.cpu().numpy(), O.dense_linear(xf.cpu().numpy().astype(np.float64), Wd, b))
        assert rel < 2e-5, (seed, N, K, fp16, M, rel)


@pytest.mark.parametrize("seed", range(8))
def test_random_column_group_layer(seed):
    """groupsize 128 / 256 (per-(row, group) levels): GEMV passes up to 11 tokens, dense workspace above"""
    u = synth.uniform01(4, 2000 + seed, 1)
    gs = 128 if seed % 2 else 256
    N, K = int(1 + u[0] * 120), gs * int(1 + u[1] * 6)
    W = synth.llm_weight(N, K, seed=50 + seed, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.85, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    fp16 = seed % 3 == 0
    Wd = r["W_fq"].astype(np.float16).astype(np.float32) if fp16 else r["W_fq"]
    layer = Q.PBLinear.from_dense(torch.from_numpy(Wd).half() if fp16 else torch.from_numpy(Wd), None, torch.from_numpy(mask), gs,
                                  r["hscale"], r["hzero"]).to(DEV)
    assert layer.packed.G == K // gs
    np.testing.assert_array_equal(layer.weight.float().cpu().numpy(), Wd)
    for M in (1, 4, 7, 11, 12, 40):
        x = synth.activations((M, K), seed, M)
        y = layer(T(x))
        rel, ratio = O.parity_errors(y.float().cpu().numpy(), O.dense_linear(x, Wd))
        assert rel < 1e-3 and ratio < 1.0, (seed, N, K, gs, M, rel, ratio)
