"""-m gpu: the GEMM-regime kernel pbl_gemm_f16_ex (csrc/pbl_gemm_big.hip; more than 32 rows of x: prefill, the reference's
perplexity loops gptq_pb/eval_ppl_utils.py:55-64, evaluate.py:126-145 call every nn.Linear with seq 2048 rows) on every layer
kind it takes: fp16-checkpoint and fp32-grid layers, column groups, fp16 and fp32 results, bias, exceptions, ragged N / K / M,
the BASELINE configs[2] layers (hessian salients: many short chunks, whole columns salient) at 11008-wide shapes -- against the
float64 oracle on the weights a dense fp16 copy of the layer holds, and against the library GEMM on pbl_unpack_dev's output
(same operands, different summation order).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pb_oracle as O
from oracle import pb_image_ref as IMG
from pb_llm_amd import _lib, synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import pack_dense
from cfg_shapes import hessian_layer
from op_trace import LIBRARY_GEMM_OPS, called_ops

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


def sample_rows(N, n=160):
    return np.unique(np.concatenate([np.arange(0, N, max(1, N // n)), [N - 1, max(N - 17, 0), min(127, N - 1), min(128, N - 1)]]))


def rtn_layer(N, K, gs, seed, low_frac, fp16, exceptions=0, metric="magnitude"):
    """RTN partially-binarized weight (per-(row, group) levels), packed; fp16: as an fp16 checkpoint holds it
    (gptq_pb/gptq.py:182).  Returns (packed, dense fp32 weight the blob unpacks to)."""
    if metric == "hessian":
        W, mask, r = hessian_layer(N, K, low_frac, seed)
    else:
        W = synth.llm_weight(N, K, seed=seed, heavy_tail=True)
        mask = O.ptq_low_mask(W, low_frac, "magnitude", None, gs)
        r = O.ptq_rtn(W, mask, 8, gs)
    G = 1 if gs == -1 else K // gs
    hi = (r["scale"] + r["mean"]).reshape(G, N).T
    lo = (-r["scale"] + r["mean"]).reshape(G, N).T
    Wd = r["W_fq"].astype(np.float32).copy()
    if fp16:
        Wd = Wd.astype(np.float16).astype(np.float32)
        hi, lo = hi.astype(np.float16).astype(np.float32), lo.astype(np.float16).astype(np.float32)
    rng = np.random.default_rng(seed)
    for _ in range(exceptions):                      # off-grid values anywhere
        Wd[rng.integers(0, N), rng.integers(0, K)] = np.float32(np.float16(rng.standard_normal()))
    p = pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=fp16)
    assert p.G == G and bool(p.flags & _lib.PBL_FLAG_SAL_F16) == fp16
    return p, Wd


CASES = [
    # N, K, M, groupsize, low_frac, fp16 checkpoint, exceptions, metric
    (256, 512, 33, -1, 0.9, True, 2, "magnitude"),
    (130, 1288, 300, -1, 0.8, True, 3, "magnitude"),            # ragged N, K % 64 = 8, ragged M
    (130, 1288, 300, -1, 0.8, False, 3, "magnitude"),           # ... fp32 grid: weights rounded to fp16 in the tile
    (72, 200, 40, -1, 0.9, False, 0, "magnitude"),              # K < one slab, K % 64 = 8
    (1000, 4096, 257, 128, 0.9, True, 4, "magnitude"),          # column groups of 128: levels change every half slab
    (272, 2048, 129, 256, 0.85, False, 2, "magnitude"),         # groups of 256, fp32 grid
    (144, 1536, 64, 512, 0.9, True, 0, "magnitude"),
    (4096, 4096, 2048, -1, 0.95, True, 0, "magnitude"),         # the configs[2] token count on the headline shape
    (11008, 4096, 512, -1, 0.95, True, 0, "hessian"),           # configs[2] gate/up: hessian salients
    (4096, 11008, 512, -1, 0.95, True, 0, "hessian"),           # configs[2] down: K = 86 half slabs
    (4096, 11008, 300, 128, 0.95, False, 0, "magnitude"),       # groupsize 128, fp32 grid, 11008 wide
    (48, 16512, 300, -1, 0.9, True, 2, "magnitude"),            # 129 half slabs: too wide for the workspace's range registers
]


@pytest.mark.parametrize("N,K,M,gs,lf,fp16,exc,metric", CASES)
def test_gemm_regime_kernel(N, K, M, gs, lf, fp16, exc, metric):
    p, Wd = rtn_layer(N, K, gs, seed=N + K + M, low_frac=lf, fp16=fp16, exceptions=exc, metric=metric)
    assert p.nexc >= exc // 2 and Q.fused_gemm_ok(p)
    pd = p.to(DEV)
    W16 = Wd.astype(np.float16).astype(np.float32)               # what a dense fp16 copy of the layer holds
    b = synth.normal((N,), 2, 3, 0.1)
    x = synth.activations((M, K), N + 3, 21)
    xt = T(x)
    rows = np.arange(N) if N * K * M < 2e9 else sample_rows(N)
    ridx = torch.from_numpy(rows).to(DEV)
    y = Q.fused_gemm_forward(pd, T(b), xt)
    assert y.shape == (M, N) and y.dtype == torch.float16
    assert_parity(y[:, ridx], O.dense_linear(x, W16[rows], b[rows]))
    y32 = Q.fused_gemm_forward(pd, None, xt, out_f32=True)
    assert y32.dtype == torch.float32
    assert_parity(y32[:, ridx], O.dense_linear(x, W16[rows]), 3e-4)       # fp32 accumulation of exact fp16 products, unrounded
    assert torch.equal(y, Q.fused_gemm_forward(pd, T(b), xt))             # deterministic
    # more than one token tile: the salient entries were decoded once per call into the workspace; inside the kernel otherwise
    assert (_lib.lib().pbl_gemm_workspace_bytes(C.byref(pd.layer_struct(None)), M) > 0) == (M > 256 and K <= 16256)
    assert torch.equal(y, Q.fused_gemm_forward(pd, T(b), xt, workspace=False))
    assert torch.equal(y32, Q.fused_gemm_forward(pd, None, xt, out_f32=True, workspace=False))
    # the two halves of the call: a list built once (it does not depend on x or M), then the GEMM over it -- any M
    lst = Q.gemm_list(pd)
    assert (lst is not None) == (K <= 16256)
    if lst is not None:
        assert torch.equal(y, Q.fused_gemm_forward(pd, T(b), xt, prepared=lst))
        assert torch.equal(y32[:33], Q.fused_gemm_forward(pd, None, xt[:33].contiguous(), out_f32=True, prepared=lst))
    # round 4: the kernel over the layer's GEMM image (pbl_gemm_f16_image) -- what the prefill path runs by default -- agrees with
    # the round-3 kernel bit for bit (same fp16 tiles, same k order per accumulator), for any M
    img = Q.gemm_image(pd)
    mx = int(pd.layer_struct(None).max_nch)                      # (no image: a slot with more than 448 entries, or more than 127 half slabs)
    assert img is not None or K > 16256 or lf <= 0.85 or metric == "hessian", mx
    if img is not None:
        assert torch.equal(y, Q.fused_gemm_forward(pd, T(b), xt, image=img))
        assert torch.equal(y32, Q.fused_gemm_forward(pd, None, xt, out_f32=True, image=img))
        assert torch.equal(y32[:33], Q.fused_gemm_forward(pd, None, xt[:33].contiguous(), out_f32=True, image=img))
        # ... and the small-batch kernel over the same image (<= 32 rows: pbl_gemm_small_image_ws; K split over the grid, partial
        # tiles added in split order): the oracle's numbers within the same tolerance, the same bits run after run
        for ms in (1, 8, 32, 33, min(M, 64)):                     # (33 - 64 rows: two blocks of 32 rows of x, the image still read once)
            xs = xt[:ms].contiguous()
            ys = Q.small_image_forward(pd, T(b), xs, img)
            assert ys.shape == (ms, N) and ys.dtype == torch.float16
            assert_parity(ys[:, ridx], O.dense_linear(x[:ms], W16[rows], b[rows]))
            ys32 = Q.small_image_forward(pd, None, xs, img, out_f32=True)
            assert_parity(ys32[:, ridx], O.dense_linear(x[:ms], W16[rows]), 3e-4)
            assert torch.equal(ys, Q.small_image_forward(pd, T(b), xs, img)) and torch.equal(ys32, Q.small_image_forward(pd, None, xs, img, out_f32=True))
        # the image itself, decoded by the independent numpy reader (oracle/pb_image_ref.py): exactly the fp16 weights a dense
        # fp16 copy of the layer holds (for an fp32-grid layer: its values rounded to fp16), every padding word idempotent
        if N * K <= 4096 * 4096:
            Wimg, st = IMG.decode(img.data.cpu().numpy())
            np.testing.assert_array_equal(Wimg.astype(np.float32), W16)
            assert st["bytes"] == img.data.numel() and sum(st["slots_by_kib"]) == ((N + 15) // 16) * ((K + 127) // 128)
        # one K split (no workspace): the same product
        lay_s = pd.layer_struct(None)
        y1 = torch.empty(8, N, dtype=torch.float32, device=DEV)
        x8 = xt[:8].contiguous()
        _lib.check(_lib.lib().pbl_gemm_small_image_ws(C.byref(lay_s), x8.data_ptr(), y1.data_ptr(), 8, 1, img.data.data_ptr(), img.data.numel(),
                                                      img.geom, None, 0, torch.cuda.current_stream().cuda_stream), "gemm_small_image")
        assert_parity(y1[:, ridx], O.dense_linear(x[:8], W16[rows]), 3e-4)
    # the library backend on the unpacked layer: same operands, different summation order
    Wdev = Q.unpack_on_device(pd, torch.float16)
    np.testing.assert_array_equal(Wdev.float().cpu().numpy()[rows], W16[rows])
    y_lib = torch.nn.functional.linear(xt, Wdev, T(b).half())
    assert_parity(y, y_lib.float().cpu().numpy().astype(np.float64), 2e-3)


def test_module_routes_the_gemm_regime_to_the_fused_kernel():
    """PBLinear above 32 rows with GEMM_BACKEND = "fused": the hand-written kernel (fp16 activations), the library backend for
    bf16 / fp32 activations; the two backends agree; a leading batch dimension and a non-contiguous input are handled."""
    p, Wd = rtn_layer(512, 1024, -1, seed=5, low_frac=0.9, fp16=True, exceptions=1)
    layer = Q.PBLinear(p.to(DEV), T(synth.normal((512,), 3, 3, 0.1)))
    x = synth.activations((3, 40, 1024), 8, 21)
    xt = T(x)
    old = Q.GEMM_BACKEND
    try:
        Q.GEMM_BACKEND = "fused"
        y = layer(xt)
        ref = O.dense_linear(x.reshape(-1, 1024), Wd, layer.pbl_bias.cpu().numpy()).reshape(3, 40, 512)
        assert y.shape == (3, 40, 512) and y.dtype == torch.float16
        assert_parity(y, ref)
        assert torch.equal(y.reshape(120, 512), Q.fused_gemm_forward(layer.packed, layer.pbl_bias, xt.reshape(120, 1024)))
        xs = T(synth.activations((120, 2048), 9, 21))[:, ::2]               # strided view
        assert_parity(layer(xs), O.dense_linear(xs.cpu().numpy(), Wd, layer.pbl_bias.cpu().numpy()))
        Q.GEMM_BACKEND = "library"
        y_lib = layer(xt)
    finally:
        Q.GEMM_BACKEND = old
    assert_parity(y, y_lib.float().cpu().numpy().astype(np.float64), 2e-3)
    # bf16 activations (round 5): the same hand-written kernels on the per-token-scaled fp16 copy, scale + bf16 cast in the GEMM's
    # epilogue; the result is rounded to bf16 (8 significand bits).  fp32 activations: two fp16 terms through the same kernel.
    ops = called_ops(lambda: layer(xt.bfloat16()))
    assert not (ops & LIBRARY_GEMM_OPS), ops
    yb = layer(xt.bfloat16())
    refb = O.dense_linear(xt.bfloat16().float().cpu().numpy().reshape(-1, 1024), Wd, layer.pbl_bias.cpu().numpy()).reshape(3, 40, 512)
    assert yb.dtype == torch.bfloat16 and O.parity_errors(yb.float().cpu().numpy(), refb)[0] < 1e-2
    assert torch.equal(yb, layer(xt.bfloat16()))
    yf = layer(xt.float())
    assert yf.dtype == torch.float32
    assert_parity(yf, ref, 1e-3)
    assert not (called_ops(lambda: layer(xt.float())) & LIBRARY_GEMM_OPS)


def test_gemm_regime_properties_at_full_size():
    """size-independent properties on the headline shape at seq 2048: linearity in x (exact for power-of-two scaling),
    token rows independent of their neighbours (a batch is its rows), zero rows give the bias."""
    N = K = 4096
    p, Wd = rtn_layer(N, K, -1, seed=41, low_frac=0.9, fp16=True)
    pd = p.to(DEV)
    b = T(synth.normal((N,), 5, 3, 0.1))
    x = T(synth.activations((2048, K), 12, 21))
    y = Q.fused_gemm_forward(pd, None, x, out_f32=True)
    assert torch.equal(Q.fused_gemm_forward(pd, None, x * 2, out_f32=True), y * 2)
    sub = Q.fused_gemm_forward(pd, None, x[700:1000].contiguous(), out_f32=True)       # other tile positions, ragged count
    assert torch.equal(sub, y[700:1000])
    z = Q.fused_gemm_forward(pd, b, torch.zeros(64, K, dtype=torch.float16, device=DEV), out_f32=True)
    assert torch.equal(z, b.expand(64, N))


def test_kept_image_and_kept_list_follow_the_blob():
    """the fused backend keeps a layer's GEMM image (quant.GEMM_KEEP_IMAGE, the default) -- or, with it off, its salient list
    (quant.GEMM_KEEP_LIST) -- between calls (perplexity loops): same results bit for bit as the per-call path, built once,
    rebuilt when the blob is written in place (its version counter moves); a call from another stream waits for the build."""
    p, Wd = rtn_layer(512, 1024, -1, seed=7, low_frac=0.9, fp16=True, exceptions=1)
    layer = Q.PBLinear(p.to(DEV), T(synth.normal((512,), 3, 3, 0.1)))
    x = T(synth.activations((300, 1024), 8, 21))
    old = (Q.GEMM_BACKEND, Q.GEMM_KEEP_LIST, Q.GEMM_KEEP_IMAGE)
    try:
        Q.GEMM_BACKEND, Q.GEMM_KEEP_LIST, Q.GEMM_KEEP_IMAGE = "fused", False, False
        ref = layer(x)
        # the kept LIST is the ctypes route's option (round 3; the native operator builds the list per call when there is no image)
        run = lambda xin: Q._pb_linear_forward(layer.packed, layer.pbl_bias, xin, False, torch.float16)       # noqa: E731
        assert torch.equal(run(x), ref)
        Q.GEMM_KEEP_LIST = True
        assert torch.equal(run(x), ref)
        kept = layer.packed._gemm_list
        assert torch.equal(run(x[:40]), ref[:40]) and layer.packed._gemm_list is kept               # one list, any M
        layer.pbl_blob.add_(0)                                                                     # written in place
        assert torch.equal(run(x), ref) and layer.packed._gemm_list[0] != kept[0]
        # backend "tuned" (round 4's default) sends a shape whose tiles do not fill the chip (here 8 of 256) to the library: same
        # weights, other summation order; a chip-filling shape takes the image kernel.  "auto" (the default) never does.
        Q.GEMM_BACKEND, Q.GEMM_KEEP_LIST, Q.GEMM_KEEP_IMAGE = "tuned", False, True
        assert not Q._image_fills_the_chip(512, 300, DEV) and Q._image_fills_the_chip(4096, 2048, DEV) and not Q._image_fills_the_chip(11008, 2048, DEV)
        assert_parity(layer(x), ref.float().cpu().numpy().astype(np.float64), 2e-3)
        assert getattr(layer.packed, "_gemm_image", None) is None
        assert called_ops(lambda: layer(x)) & LIBRARY_GEMM_OPS
        for backend in ("auto",):
            Q.GEMM_BACKEND, Q.GEMM_KEEP_LIST, Q.GEMM_KEEP_IMAGE = backend, False, True
            assert torch.equal(layer(x), ref)
            kimg = layer.packed._gemm_image
            assert kimg[1] is not None and kimg[1].max_slot_kib >= 1
            assert torch.equal(layer(x[:70]), ref[:70]) and layer.packed._gemm_image is kimg      # one image, any M
            # ... and up to 64 rows the small-batch kernel over the same image (another summation order: the K split)
            y40 = layer(x[:40])
            assert torch.equal(y40, Q.small_image_forward(layer.packed, layer.pbl_bias, x[:40].contiguous(), kimg[1])) and layer.packed._gemm_image is kimg
            assert_parity(y40, ref[:40].float().cpu().numpy().astype(np.float64), 2e-3)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ys = layer(x)
            torch.cuda.current_stream().wait_stream(side)
            assert torch.equal(ys, ref)
            layer.pbl_blob.add_(0)
            assert torch.equal(layer(x), ref) and layer.packed._gemm_image[0] != kimg[0]
    finally:
        Q.GEMM_BACKEND, Q.GEMM_KEEP_LIST, Q.GEMM_KEEP_IMAGE = old


def test_gemm_regime_fully_binarized_layer_in_list_mode():
    """a layer without ANY salient entry or exception (BinaryLinear / XnorBinaryLinear: max_nch = max_nexc = 0) on more than one
    token tile: the salient list has no words, and its capacity must still be a valid region for the producers' clamped,
    unconditional requests (round 3 computed cap - 1 = 0xFFFFFFFF and read past the workspace)"""
    N, K, M = 256, 1024, 600
    rng = np.random.default_rng(5)
    a = np.abs(synth.llm_weight(N, 1, seed=9)[:, 0]).astype(np.float32) + 0.01
    a = a.astype(np.float16).astype(np.float32)
    Wd = np.where(rng.random((N, K)) < 0.5, a[:, None], -a[:, None]).astype(np.float32)
    p = pack_dense(Wd, a, -a, np.ones(N, np.float32), np.zeros(N, np.float32), None, sal_f16=True)
    assert p.nnz == 0 and p.nexc == 0 and p.max_nch == 0 and Q.fused_gemm_ok(p)
    pd = p.to(DEV)
    x = synth.activations((M, K), 11, 21)
    y = Q.fused_gemm_forward(pd, None, T(x))
    assert_parity(y, O.dense_linear(x, Wd))
    lst = Q.gemm_list(pd)
    assert lst is not None and lst.numel() > 0
    assert torch.equal(y, Q.fused_gemm_forward(pd, None, T(x), prepared=lst))
    assert torch.equal(y, Q.fused_gemm_forward(pd, None, T(x), workspace=False))
    img = Q.gemm_image(pd)
    assert img is not None and img.max_slot_kib == 1
    assert torch.equal(y, Q.fused_gemm_forward(pd, None, T(x), image=img))


def test_image_kernel_repeatedly_against_the_round3_kernel_at_full_size():
    """the image kernel's waves hand stages over with counted waits and one barrier per 64-column step; a count that is too lax
    shows up as a RARE wrong tile (round 4: 2 of 256 workgroups in one of two layers, found by the config-3 test).  The full-size
    hessian down_proj layer, 2048 rows, 25 launches back to back: every one bit-identical to the round-3 kernel's result."""
    from cfg_shapes import hessian_layer
    N, K, M = 4096, 11008, 2048
    W, mask, r = hessian_layer(N, K, 0.95, seed=302)
    layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    p = layer.packed
    x = T(synth.activations((M, K), 77, 21))
    img = Q.gemm_image(p)
    assert img is not None
    ref = Q.fused_gemm_forward(p, None, x)
    outs = [Q.fused_gemm_forward(p, None, x, image=img) for _ in range(25)]
    torch.cuda.synchronize()
    bad = [i for i, y in enumerate(outs) if not torch.equal(y, ref)]
    assert not bad, f"launches {bad} differ from the round-3 kernel"


@pytest.mark.parametrize("seed", range(10))
def test_small_batch_image_kernel_on_random_layers(seed):
    """the small-batch kernel over the GEMM image on random small layers: ragged N (an odd record count, fewer rows than a pair of
    records), K from one half slab to several with and without a tail, column groups, exceptions, every row count class (1, one
    block of 32, two blocks), with and without a bias / fp32 result -- against the float64 oracle on the fp16 weights"""
    rng = np.random.default_rng(1000 + seed)
    gs = int(rng.choice([-1, -1, 128, 256]))
    if gs == -1:
        K = int(rng.choice([16, 72, 128, 200, 384, 1288, 2048]))
    else:
        K = gs * int(rng.integers(1, 9))
    N = int(rng.choice([8, 16, 24, 40, 100, 130, 257, 512]))
    lf = float(rng.choice([0.5, 0.8, 0.9, 0.97]))
    p, Wd = rtn_layer(N, K, gs, seed=77 + seed, low_frac=lf, fp16=True, exceptions=int(rng.integers(0, 4)))
    pd = p.to(DEV)
    img = Q.gemm_image(pd)
    assert img is not None
    W16 = Wd.astype(np.float16).astype(np.float32)
    np.testing.assert_array_equal(IMG.decode(img.data.cpu().numpy())[0].astype(np.float32), W16)
    b = synth.normal((N,), 5, seed, 0.1)
    for M in (1, int(rng.integers(2, 32)), 32, int(rng.integers(33, 64)), 64):
        x = synth.activations((M, K), seed + M, 21)
        xt = T(x)
        y = Q.small_image_forward(pd, T(b), xt, img)
        assert y.shape == (M, N)
        assert_parity(y, O.dense_linear(x, W16, b))
        y32 = Q.small_image_forward(pd, None, xt, img, out_f32=True)
        assert_parity(y32, O.dense_linear(x, W16), 3e-4)
        assert torch.equal(y32, Q.small_image_forward(pd, None, xt, img, out_f32=True))


def test_small_batch_image_kernel_repeatedly_at_full_size():
    """the staging wave hands x tiles to the working waves through a double buffer and one barrier per half slab; a hand-over that
    is too early shows as a result that differs from launch to launch: 100 launches each at 32 and 64 rows on the configs[3] shape,
    other work (the image GEMM kernel) in between"""
    N, K = 13824, 5120
    p, Wd = rtn_layer(N, K, -1, seed=5, low_frac=0.8, fp16=True)
    pd = p.to(DEV)
    img = Q.gemm_image(pd)
    assert img is not None
    W16 = Wd.astype(np.float16).astype(np.float32)
    rows = sample_rows(N)
    ridx = torch.from_numpy(rows).to(DEV)
    xbig = T(synth.activations((512, K), 4, 21))
    for M in (32, 64):
        x = synth.activations((M, K), 9 + M, 21)
        xt = T(x)
        ref = Q.small_image_forward(pd, None, xt, img)
        assert_parity(ref[:, ridx], O.dense_linear(x, W16[rows]))
        outs = []
        for i in range(100):
            if i % 25 == 0:
                Q.fused_gemm_forward(pd, None, xbig, image=img)          # (another kernel's LDS contents and cache state in between)
            outs.append(Q.small_image_forward(pd, None, xt, img))
        torch.cuda.synchronize()
        bad = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
        assert not bad, f"M={M}: launches {bad[:10]} differ"


def _force_plan(mode, cut=0, ks=0):
    f = _lib.lib().pbl_debug_force_gemm_plan
    f.restype, f.argtypes = None, [C.c_int, C.c_int, C.c_int]
    f(mode, cut, ks)


@pytest.mark.parametrize("N,K,M,bias", [(512, 1024, 600, True), (520, 1288, 700, False)])
def test_gemm_image_k_split_tail_every_plan(N, K, M, bias):
    """pbl_gemm_f16_image_ws (round 5): the launch cut into a full part and a K-split tail, every plan shape forced on a small layer
    (pbl_debug_force_gemm_plan) -- token tail / row tail, with and without a full part, 2 - 4 splits, K % 64 != 0 and a ragged last row
    tile -- for fp16, fp32 and bf16 (+ per-token scale) results with and without bias: the full part's tiles keep the one-launch bits,
    the tail agrees within the parity tolerance, against the float64 oracle, repeatable."""
    p, Wd = rtn_layer(N, K, -1, seed=N + 3, low_frac=0.9, fp16=True, exceptions=1)
    pd = p.to(DEV)
    b = T(synth.normal((N,), 4, 3, 0.1)) if bias else None
    x = synth.activations((M, K), 5, 21)
    xt = T(x)
    ref = O.dense_linear(x, Wd, None if b is None else b.cpu().numpy())
    img = Q.gemm_image(pd)
    assert img is not None
    y0 = Q.fused_gemm_forward(pd, b, xt, image=img)                        # one launch
    y0f = Q.fused_gemm_forward(pd, b, xt, out_f32=True, image=img)
    xb = (xt.float() * 3.0e4).bfloat16()                                   # beyond fp16's range: scaled per token
    xh, tsc = Q.act_bf16_prepare(xb)
    yb0 = Q.fused_gemm_forward(pd, b, xh, image=img, tok_scale=tsc)
    assert yb0.dtype == torch.bfloat16
    RT, TT = (N + 127) // 128, (M + 255) // 256
    try:
        for mode, cut, ks in ((1, TT - 1, 2), (1, 1, 3), (1, 0, 2), (2, RT - 1, 2), (2, 2, 4), (2, 0, 2)):
            _force_plan(mode, cut, ks)
            plan = (C.c_uint64 * 6)()
            lay = pd.layer_struct(None)
            assert _lib.lib().pbl_gemm_image_plan(C.byref(lay), M, plan) == 0 and plan[0] == mode and plan[1] == cut, (mode, cut, list(plan))
            y = Q.fused_gemm_forward(pd, b, xt, image=img, split_k=True)
            assert_parity(y, ref)
            assert_parity(y, y0.float().cpu().numpy().astype(np.float64), 2e-3)
            full = (slice(0, cut * 256), slice(None)) if mode == 1 else (slice(None), slice(0, cut * 128))
            tail = (slice(cut * 256, None), slice(None)) if mode == 1 else (slice(None), slice(cut * 128, None))
            assert torch.equal(y[full], y0[full]), (mode, cut, ks)                       # the full part: the same launch geometry, the same bits
            assert not torch.equal(y[tail], y0[tail])                                    # (the tail did take another summation order)
            assert torch.equal(y, Q.fused_gemm_forward(pd, b, xt, image=img, split_k=True))
            yf = Q.fused_gemm_forward(pd, b, xt, out_f32=True, image=img, split_k=True)
            assert yf.dtype == torch.float32 and torch.equal(yf[full], y0f[full])
            assert_parity(yf, ref, 2e-4)
            yb = Q.fused_gemm_forward(pd, b, xh, image=img, tok_scale=tsc, split_k=True)
            assert yb.dtype == torch.bfloat16 and torch.equal(yb[full], yb0[full])
            assert O.parity_errors(yb.float().cpu().numpy(), yb0.float().cpu().numpy().astype(np.float64))[0] < 1e-2
    finally:
        _force_plan(-1)
    # (back on the cost model: whatever it picks for this small layer, the module route agrees with the one-launch result)
    layer = Q.PBLinear(pd, b)
    assert_parity(layer(xt), y0.float().cpu().numpy().astype(np.float64), 2e-3)


def test_gemm_image_short_prompt_is_split_along_k():
    """a 300-token prompt on a 4096 x 4096 layer is 64 tiles of 128 x 256 on 256 CUs: the cost model splits every tile along K (token
    tail with an empty full part) -- through the module (the shipped default), against the oracle and the one-launch result"""
    p, Wd = rtn_layer(4096, 4096, -1, seed=77, low_frac=0.95, fp16=True)
    layer = Q.PBLinear(p.to(DEV), None)
    plan = (C.c_uint64 * 6)()
    lay = layer.packed.layer_struct(None)
    assert _lib.lib().pbl_gemm_image_plan(C.byref(lay), 300, plan) == 0
    assert plan[0] == 1 and plan[1] == 0 and plan[2] >= 2 and plan[4] == 300 and plan[5] == 4096, list(plan)
    assert _lib.lib().pbl_gemm_image_plan(C.byref(lay), 2048, plan) == 0 and plan[0] == 0       # 256 tiles: exactly one round
    x = synth.activations((300, 4096), 9, 21)
    xt = T(x)
    assert Q.GEMM_SPLIT_K and Q.GEMM_BACKEND == "auto"
    y = layer(xt)
    rows = sample_rows(4096)
    assert_parity(y[:, torch.from_numpy(rows).to(DEV)], O.dense_linear(x, Wd[rows]))
    assert torch.equal(y, layer(xt))
    y1 = Q.fused_gemm_forward(layer.packed, None, xt, image=layer.packed._gemm_image[1])
    assert not torch.equal(y, y1)
    assert_parity(y, y1.float().cpu().numpy().astype(np.float64), 2e-3)
    yb = layer(xt.bfloat16())
    assert yb.dtype == torch.bfloat16 and O.parity_errors(yb[:, torch.from_numpy(rows).to(DEV)].float().cpu().numpy(),
                                                          O.dense_linear(xt.bfloat16().float().cpu().numpy(), Wd[rows]))[0] < 1e-2


_full = {}


def _full_layer():
    """one real shape for the full-tensor checks: 4096 x 4096, low_frac 0.95 hessian salients (the configs[2] q_proj kind), fp16 checkpoint"""
    if "l" not in _full:
        p, Wd = rtn_layer(4096, 4096, -1, seed=311, low_frac=0.95, fp16=True, metric="hessian")
        pd = p.to(DEV)
        _full["l"] = (pd, Wd, Q.gemm_image(pd))
    return _full["l"]


@pytest.mark.parametrize("family", ["gemm_one_launch", "gemm_k_split_tail", "small_batch_32", "small_batch_64"])
def test_full_tensor_against_the_float64_oracle_per_kernel_family(family):
    """the real-shape GEMM-regime parity tests compare a SAMPLE of ~190 output rows with the oracle (the float64
    GEMM is the slow part) plus a whole-tensor self-comparison -- a wrong tile that misses the sampled rows is only caught by the
    self-comparison, the class of bug round 4 shipped (`vmcnt(13)`).  Here ALL N rows of a 4096 x 4096 layer are compared with
    O.dense_linear, one case per kernel family: pbl_gemm_img_kernel<0> as one launch at 2048 rows, the K-split tail
    (pbl_gemm_img_kernel<1> partial tiles + img_reduce_kernel; forced row-tail plan with a full part), and the small-batch kernel
    at 32 and 64 rows (both token-block instantiations)."""
    pd, Wd, img = _full_layer()
    assert img is not None
    M = {"gemm_one_launch": 2048, "gemm_k_split_tail": 2048, "small_batch_32": 32, "small_batch_64": 64}[family]
    x = synth.activations((M, 4096), 41, 21)
    xt = T(x)
    ref = O.dense_linear(x, Wd)                                            # [M, 4096] float64: every row of the layer
    if family == "gemm_one_launch":
        y = Q.fused_gemm_forward(pd, None, xt, image=img)
    elif family == "gemm_k_split_tail":
        try:
            _force_plan(2, 16, 4)                                          # row tiles [0, 16) in one piece, [16, 32) split 4 ways along K
            y = Q.fused_gemm_forward(pd, None, xt, image=img, split_k=True)
        finally:
            _force_plan(-1)
        y1 = Q.fused_gemm_forward(pd, None, xt, image=img)
        assert torch.equal(y[:, :2048], y1[:, :2048]) and not torch.equal(y[:, 2048:], y1[:, 2048:])
    else:
        y = Q.small_image_forward(pd, None, xt, img)
    assert y.shape == (M, 4096) and y.dtype == torch.float16
    assert_parity(y, ref)
    # and nothing is merely "within tolerance on average": the worst element of EVERY 128-row x 256-token tile is inside the bar
    err = np.abs(y.float().cpu().numpy().astype(np.float64) - ref)
    bar = 1e-3 * np.abs(ref).max()
    tiles = err.reshape(-1, min(M, 256), 32, 128).max(axis=(1, 3)) if M >= 256 else err.reshape(1, M, 32, 128).max(axis=(1, 3))
    assert (tiles < bar).all(), np.argwhere(tiles >= bar)[:8]


def _force_small_plan(geo, ks=0):
    f = _lib.lib().pbl_debug_set_small_image_plan
    f.restype, f.argtypes = None, [C.c_int, C.c_int]
    f(geo, ks)


@pytest.mark.parametrize("seed", range(8))
def test_small_batch_geometries_with_k_phases_on_random_layers(seed):
    """round 6: the small-batch kernel's workgroup is (RP row pairs) x (KQ K phases) -- geometry 1 = 2 x 4 (whole-K workgroups: no
    split across workgroups, no workspace, one launch) -- the phases' accumulators added through LDS in phase order.  The geometry forced (pbl_debug_set_small_image_plan) on random layers: ragged N / odd record counts, K from one half slab
    (fewer half slabs than phases) to many with and without a tail, column groups, exceptions; with and without an additional split
    across workgroups; 1 - 32 rows; against the float64 oracle, repeatable, and the same numbers as geometry 0 within fp32 summation
    order."""
    rng = np.random.default_rng(2000 + seed)
    gs = int(rng.choice([-1, -1, 128, 256]))
    K = int(rng.choice([16, 128, 200, 384, 1288, 2048, 3200])) if gs == -1 else gs * int(rng.integers(1, 12))
    N = int(rng.choice([8, 24, 40, 100, 130, 257, 512]))
    lf = float(rng.choice([0.5, 0.8, 0.9, 0.97]))
    p, Wd = rtn_layer(N, K, gs, seed=177 + seed, low_frac=lf, fp16=True, exceptions=int(rng.integers(0, 4)))
    pd = p.to(DEV)
    img = Q.gemm_image(pd)
    assert img is not None
    W16 = Wd.astype(np.float16).astype(np.float32)
    b = synth.normal((N,), 5, seed, 0.1)
    NH = (K + 127) // 128
    try:
        for M in (1, int(rng.integers(2, 32)), 32):
            x = synth.activations((M, K), seed + M, 21)
            xt = T(x)
            ref, ref_nb = O.dense_linear(x, W16, b), O.dense_linear(x, W16)
            _force_small_plan(0)
            y0 = Q.small_image_forward(pd, None, xt, img, out_f32=True)
            for geo, ks in ((1, 1), (1, 2), (1, 3), (1, NH)):
                _force_small_plan(geo, ks)
                y = Q.small_image_forward(pd, T(b), xt, img)
                assert y.shape == (M, N)
                assert_parity(y, ref)
                y32 = Q.small_image_forward(pd, None, xt, img, out_f32=True)
                assert_parity(y32, ref_nb, 3e-4)
                assert torch.equal(y32, Q.small_image_forward(pd, None, xt, img, out_f32=True)), (geo, ks, M)
                assert_parity(y32, y0.cpu().numpy().astype(np.float64), 1e-4)
    finally:
        _force_small_plan(-1)


@pytest.mark.parametrize("N,K,geo", [(13824, 5120, 1), (5120, 13824, 0)])
def test_small_batch_default_geometry_on_the_config4_shapes(N, K, geo):
    """BASELINE configs[3] out of the box (round 6): 13824 x 5120 at 32 rows runs as ONE launch -- whole-K workgroups, no workspace
    (pbl_gemm_small_image_workspace_bytes == 0) -- and 5120 x 13824 (160 row pairs) keeps the split across workgroups; ALL rows
    against the float64 oracle, 50 launches bit-equal (the x double buffer / phase reduction hand-overs), 1 and 17 rows as well."""
    p, Wd = rtn_layer(N, K, -1, seed=5 + geo, low_frac=0.8, fp16=True)
    pd = p.to(DEV)
    img = Q.gemm_image(pd)
    assert img is not None
    W16 = Wd.astype(np.float16).astype(np.float32)
    lay = pd.layer_struct(None)
    wsb = _lib.lib().pbl_gemm_small_image_workspace_bytes
    wsb.restype = C.c_size_t
    if geo == 1:
        assert wsb(C.byref(lay), 32) == 0                                   # one launch, nothing to add up afterwards
    else:
        assert wsb(C.byref(lay), 32) > 0
    for M in (32, 17, 1):
        x = synth.activations((M, K), 9 + M, 21)
        xt = T(x)
        y = Q.small_image_forward(pd, None, xt, img)
        assert_parity(y, O.dense_linear(x, W16))
        for _ in range(50 if M == 32 else 3):
            assert torch.equal(y, Q.small_image_forward(pd, None, xt, img))


@pytest.mark.parametrize("N,K,M,gs", [(512, 1024, 600, -1), (520, 1288, 700, -1), (256, 1024, 257, 256), (384, 200, 70, -1), (4096, 4096, 2048, -1)])
def test_gemm_image_kernel_with_x_as_fragment_major_copy(N, K, M, gs):
    """round 6: pbl_x_to_fragments + pbl_gemm_f16_image_xf -- the MFMA waves load their B fragments straight from a fragment-major copy
    of x, no x tile goes through LDS.  Every accumulator sums the same products in the same order: BIT-IDENTICAL to the round-4 kernel
    (one launch and every forced K-split plan), for fp16 / fp32 / bf16 (+ per-token scale) results, K % 64 != 0, ragged M and N, column
    groups; the copy itself against a numpy restatement of its layout."""
    p, Wd = rtn_layer(N, K, gs, seed=N + K, low_frac=0.9 if N < 4096 else 0.95, fp16=True, exceptions=1 if N < 4096 else 0)
    pd = p.to(DEV)
    b = T(synth.normal((N,), 4, 3, 0.1))
    x = synth.activations((M, K), 5, 21)
    xt = T(x)
    img = Q.gemm_image(pd)
    assert img is not None
    xf = Q.x_fragments(xt)
    # the layout: [token block of 32][k-step of 16 columns][lane: token l & 31, columns 8 (l >> 5) .. + 7][8 halves], zero padded
    kp = (K + 63) // 64 * 64 + 64
    Mp = (M + 255) // 256 * 256
    assert xf.numel() == Mp * kp * 2
    xp = np.zeros((Mp, kp), np.float16)
    xp[:M, :K] = x
    want = xp.reshape(Mp // 32, 32, kp // 16, 2, 8).transpose(0, 2, 3, 1, 4)
    np.testing.assert_array_equal(xf.view(torch.float16).cpu().numpy().reshape(want.shape), want)
    y0 = Q.fused_gemm_forward(pd, b, xt, image=img)
    y1 = Q.fused_gemm_forward(pd, b, xt, image=img, x_frag=xf)
    assert torch.equal(y0, y1)
    if N < 4096:
        assert_parity(y1, O.dense_linear(x, Wd, b.cpu().numpy()))
    assert torch.equal(Q.fused_gemm_forward(pd, None, xt, out_f32=True, image=img), Q.fused_gemm_forward(pd, None, xt, out_f32=True, image=img, x_frag=xf))
    xb = (xt.float() * 3.0e4).bfloat16()
    xh, tsc = Q.act_bf16_prepare(xb)
    assert torch.equal(Q.fused_gemm_forward(pd, b, xh, image=img, tok_scale=tsc), Q.fused_gemm_forward(pd, b, xh, image=img, tok_scale=tsc, x_frag=True))
    RT, TT = (N + 127) // 128, (M + 255) // 256
    NH = (K + 127) // 128
    if NH >= 4:
        try:
            for mode, cut, ks in ((1, max(TT - 1, 0), 2), (2, max(RT - 1, 0), 2), (1, 0, 2)):
                _force_plan(mode, cut, ks)
                ya = Q.fused_gemm_forward(pd, b, xt, image=img, split_k=True)
                yb = Q.fused_gemm_forward(pd, b, xt, image=img, split_k=True, x_frag=xf)
                assert torch.equal(ya, yb), (mode, cut, ks)
        finally:
            _force_plan(-1)
    for _ in range(5):
        assert torch.equal(y1, Q.fused_gemm_forward(pd, b, xt, image=img, x_frag=xf))


def test_modules_share_one_fragment_copy_per_activation_tensor(monkeypatch):
    """round 6, the module path of x as a fragment-major copy: projections called with the SAME activation tensor (q / k / v, gate / up:
    gptq_pb/eval_ppl_utils.py:55-64) share one copy -- kept per device, keyed on the tensor object, its version counter, the stream and
    the shape --; an in-place change of the tensor, another tensor or another shape makes a new one; fp16 / bf16 / fp32 activations
    give the bits of the LDS-staged kernel (GEMM_X_FRAGMENTS = False), with a leading batch dimension and a strided view; the image
    is the layers' only device copy as well."""
    layers = []
    for seed, N in ((5, 512), (6, 384), (7, 640)):
        p, _ = rtn_layer(N, 1024, -1, seed=seed, low_frac=0.9, fp16=True, exceptions=1)
        layers.append(Q.PBLinear(p.to(DEV), T(synth.normal((N,), seed, 3, 0.1))))
    x = T(synth.activations((2, 150, 1024), 8, 21))
    assert Q.GEMM_X_FRAGMENTS
    made = []
    real = Q.x_fragments

    def counting(x2, cache_key=None):
        before = Q._XF_CACHE.get(x2.device.index)
        out = real(x2, cache_key=cache_key)
        if Q._XF_CACHE.get(x2.device.index) is not before or cache_key is None:
            made.append(tuple(x2.shape))
        return out

    monkeypatch.setattr(Q, "x_fragments", counting)
    Q.drop_x_fragments_()
    with torch.no_grad():
        ys = [l(x) for l in layers]
        assert made == [(300, 1024)]                                  # one copy for the three projections
        x.mul_(0.5)                                                   # the same object, new contents: a new copy
        ys2 = [l(x) for l in layers]
        assert made == [(300, 1024)] * 2
        for a, b in zip(ys, ys2):
            assert not torch.equal(a, b)
        xs = T(synth.activations((70, 2048), 9, 21))[:, ::2]          # a strided view (made contiguous first), another shape
        yv = layers[0](xs)
        assert made[-1] == (70, 1024) and len(made) == 3
        yb = [l(x.bfloat16()) for l in layers[:2]]
        yf = layers[0](x.float())
        monkeypatch.setattr(Q, "GEMM_X_FRAGMENTS", False)
        n = len(made)
        for l, a in zip(layers, ys2):
            assert torch.equal(l(x), a)
        assert torch.equal(layers[0](xs), yv)
        for l, a in zip(layers, yb):
            assert torch.equal(l(x.bfloat16()), a)
        assert torch.equal(layers[0](x.float()), yf)
        assert len(made) == n                                         # (switched off: no copy is made)
        monkeypatch.setattr(Q, "GEMM_X_FRAGMENTS", True)
        layers[1].release_blob_()                                     # image-only residency: the same bits from the fragment path
        assert torch.equal(layers[1](x), ys2[1])
    Q.drop_x_fragments_()
    assert not Q._XF_CACHE
