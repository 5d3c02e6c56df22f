"""-m gpu: BASELINE.json configs[2] (llama-7b linears, low_frac 0.95 hessian, M = 2048) and configs[4] (the same layers
row-sharded over 2/4/8 ranks) through the HIP path, plus the rows round 1 left at CPU-only coverage: the on-disk format
(utils.py:65-124), the Hessian-mask module (a8, quant/outlier_quantizer.py:126-143), the input gradient
of the packed forward and the staleness of the packed cache.
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch

from oracle import pb_oracle as O
from pb_llm_amd import _lib, synth
from pb_llm_amd import io as pbio
from pb_llm_amd import parallel as PP
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import PackedWeight, pack_dense
from cfg_shapes import LLAMA7B, LLAMA7B_DISTINCT, hessian_layer
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


_cache = {}


def cfg3_layer(name, lf=0.95):
    """packed llama-7b linear `name` at low_frac `lf` (0.95: configs[2]; 0.9: configs[4]) with hessian salients (fp16
    checkpoint, as gptq_pb writes it)"""
    key = name if lf == 0.95 else f"{name}@{lf}"
    if key not in _cache:
        N, K = LLAMA7B[name]
        W, mask, r = hessian_layer(N, K, lf, seed=300 + len(_cache) // 2)
        W16 = torch.from_numpy(r["W_fq"]).half()
        layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
        _cache[key] = (layer, W16, mask)
        _cache[key + "_rtn"] = r
    return _cache[key]


# ------------------------------------------------------------------------------------------- config 3
@pytest.mark.parametrize("name", list(LLAMA7B_DISTINCT))
def test_config3_llama7b_linears_hessian_m2048(name):
    """configs[2]: every decoder linear of llama-7b (seven, three distinct shapes), low_frac 0.95, hessian salients from a
    column-concentrated Hessian, seq 2048 -- the HIP path against the float64 oracle on a sample of output rows."""
    assert sorted(sum(LLAMA7B_DISTINCT.values(), ())) == sorted(LLAMA7B)          # the three shapes cover all seven linears
    layer, W16, mask = cfg3_layer(name)
    N, K = LLAMA7B[name]
    p = layer.packed
    assert (p.N, p.K) == (N, K) and p.flags & _lib.PBL_FLAG_SAL_F16
    sal_frac = 1.0 - mask.mean()
    assert abs(sal_frac - 0.05) < 2e-3 and p.nexc < 1e-3 * N * K
    # hessian saliency is column concentrated: some input channels are salient for (nearly) every row
    col_frac = (~mask).mean(0)
    assert (col_frac > 0.9).sum() >= 8 and np.median(col_frac) < 0.05
    x = synth.activations((2048, K), 77, 21)
    rows = np.unique(np.concatenate([np.arange(0, N, max(1, N // 192)), [N - 1, N - 16, 15, 16]]))
    ref = O.dense_linear(x, W16.numpy()[rows])
    assert Q.GEMM_BACKEND == "auto"                        # the shipped default: ALWAYS the hand-written kernel (round 5) ...
    xt = T(x)
    ops = called_ops(lambda: layer(xt))
    assert not (ops & LIBRARY_GEMM_OPS), (name, ops & LIBRARY_GEMM_OPS)      # ... at::linear / mm never runs, on any of the three shapes
    y_auto = layer(xt)
    assert y_auto.shape == (2048, N) and y_auto.dtype == torch.float16
    assert_parity(y_auto[:, torch.from_numpy(rows).to(DEV)], ref)
    assert layer.packed._gemm_image[1] is not None
    # bf16 activations (qat/run_qat.py:120): the same kernel on the per-token-scaled fp16 copy, scale + cast in its epilogue
    xb = xt.bfloat16()
    assert not (called_ops(lambda: layer(xb)) & LIBRARY_GEMM_OPS)
    yb = layer(xb)
    assert yb.dtype == torch.bfloat16
    refb = O.dense_linear(xb.float().cpu().numpy(), W16.numpy()[rows])
    assert O.parity_errors(yb[:, torch.from_numpy(rows).to(DEV)].float().cpu().numpy(), refb)[0] < 1e-2      # bf16 result: 8 significand bits
    old = Q.GEMM_BACKEND
    try:
        for backend in ("library", "tuned"):               # the library implementation, and round 4's routing between the two
            Q.GEMM_BACKEND = backend
            y = layer(xt)
            assert y.shape == (2048, N) and y.dtype == torch.float16
            assert_parity(y[:, torch.from_numpy(rows).to(DEV)], ref)
            assert_parity(y, y_auto.float().cpu().numpy().astype(np.float64), 2e-3)
    finally:
        Q.GEMM_BACKEND = old
    # the decode-time regimes of the same layer: GEMV (1 token) and the matrix-core kernel (32 tokens)
    for M in (1, 32):
        assert_parity(layer(T(x[:M])), O.dense_linear(x[:M], W16.numpy()))
    if name == "gate_proj":
        # 1536 rows on 11008 x 4096 are 516 tiles -- 4 more than two rounds of the chip: the launch plan cuts the last row tile off and
        # splits its 6 tiles 8 ways (4 half slabs per work item: the shortest K range a plan uses)
        plan = (C.c_uint64 * 6)()
        assert _lib.lib().pbl_gemm_image_plan(C.byref(layer.packed.layer_struct(None)), 1536, plan) == 0 and list(plan)[:4] == [2, 85, 8, 4]
        y15 = layer(T(x[:1536]))
        assert_parity(y15[:, torch.from_numpy(rows).to(DEV)], O.dense_linear(x[:1536], W16.numpy()[rows]))
        assert torch.equal(y15[:, :85 * 128], y_auto[:1536, :85 * 128]) and torch.equal(y15, layer(T(x[:1536])))


@pytest.mark.parametrize("N,K", [(5120, 5120), (13824, 5120), (5120, 13824)])
def test_llama13b_linears_m2048_default_backend_is_hand_written(N, K):
    """every llama-13b linear shape at 2048 rows (the reference's perplexity loops, gptq_pb/eval_ppl_utils.py:55-64): round 4's
    default sent all of them to unpack + library GEMM (320 / 864 tiles of 128 x 256 leave a thin last round); the shipped default
    now multiplies from the GEMM image -- no ATen GEMM operator is reached -- with fp16 and bf16 activations; against the float64
    oracle on a sample of rows and against the library backend."""
    W = synth.llm_weight(N, K, seed=N % 89)
    mask = O.ptq_low_mask(W, 0.95, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    x = synth.activations((2048, K), 78, 21)
    xt = T(x)
    rows = np.unique(np.concatenate([np.arange(0, N, max(1, N // 160)), [N - 1, N - 16, 15, 16]]))
    ref = O.dense_linear(x, W16.numpy()[rows])
    assert Q.GEMM_BACKEND == "auto" and Q.GEMM_SPLIT_K
    assert not (called_ops(lambda: layer(xt)) & LIBRARY_GEMM_OPS)
    y = layer(xt)
    assert_parity(y[:, torch.from_numpy(rows).to(DEV)], ref)
    assert torch.equal(y, layer(xt))                                       # repeatable
    # 5120-row layers are 320 tiles on 256 CUs: the plan cuts the last 8 row tiles off and splits them along K (round 5); the tiles of
    # the full part keep the one-launch bits, the tail differs by summation order only
    plan = (C.c_uint64 * 6)()
    lay = layer.packed.layer_struct(None)
    assert _lib.lib().pbl_gemm_image_plan(C.byref(lay), 2048, plan) == 0
    if N == 5120:
        assert plan[0] == 2 and plan[1] == 32 and plan[2] >= 2 and plan[5] == 1024, list(plan)
    Q.GEMM_SPLIT_K = False
    try:
        y1 = layer(xt)
    finally:
        Q.GEMM_SPLIT_K = True
    assert_parity(y, y1.float().cpu().numpy().astype(np.float64), 2e-3)
    if plan[0] == 2:
        c0 = int(plan[1]) * 128
        assert torch.equal(y[:, :c0], y1[:, :c0]) and not torch.equal(y[:, c0:], y1[:, c0:])
    elif plan[0] == 0:
        assert torch.equal(y, y1)
    xb = xt.bfloat16()
    assert not (called_ops(lambda: layer(xb)) & LIBRARY_GEMM_OPS)
    refb = O.dense_linear(xb.float().cpu().numpy(), W16.numpy()[rows])
    assert O.parity_errors(layer(xb)[:, torch.from_numpy(rows).to(DEV)].float().cpu().numpy(), refb)[0] < 1e-2
    old = Q.GEMM_BACKEND
    try:
        Q.GEMM_BACKEND = "library"
        assert called_ops(lambda: layer(xt)) & LIBRARY_GEMM_OPS                # (the probe does see the library when it runs)
        assert_parity(layer(xt), y.float().cpu().numpy().astype(np.float64), 2e-3)
    finally:
        Q.GEMM_BACKEND = old


def test_config3_llama7b_width_model_prefill_2048_and_graphed_decode():
    """configs[2] at REAL width under -m gpu (round 4 had it at toy size only): a random-init HF LlamaForCausalLM with llama-7b's
    hidden 4096 / intermediate 11008 / 32 heads (2 decoder layers and a small vocabulary so that it fits the time budget), every
    decoder Linear quantised by the oracle's restatement of gptq_pb RTN (low_frac 0.95), swapped for PBLinear
    (harness.to_pb_, utils.py:97-124's attribute replacement).  Prefill of 2048 tokens with the shipped default backend -- the
    hand-written kernel over each layer's GEMM image, all 14 linears -- and 8 graph-replayed decode steps of the fused model
    (harness.build_gemm_images_ / fuse_decode_ / GraphedForward), logits against the SAME model with dense fake-quant fp16 weights:
    what gptq_pb/eval_ppl_utils.py:8-88 and qat/eval_after_qat.py:11-33 evaluate."""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    from pb_llm_amd import harness as H
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=2048, max_position_embeddings=2048)
    model = LlamaForCausalLM(cfg).half().eval()
    model.model.layers[1].load_state_dict(model.model.layers[0].state_dict())       # (one set of seven weights to quantise on the host)
    done = {}

    def producer(name, W):
        key = name.split("layers.")[1].split(".", 1)[1]
        if key not in done:
            Wn = W.float().numpy()
            mask = O.ptq_low_mask(Wn, 0.95, "magnitude", None, -1)
            r = O.ptq_rtn(Wn, mask, 8, -1)
            done[key] = dict(W_fq=torch.from_numpy(r["W_fq"]), low_mask=torch.from_numpy(mask), hscale=r["hscale"], hzero=r["hzero"])
        return done[key]

    side = H.quantize_dense_(model, producer)
    assert len(side) == 14 and len(done) == 7
    dense = copy.deepcopy(model).to(DEV)
    pb = H.to_pb_(model, side).to(DEV)
    lins = [m for m in pb.modules() if isinstance(m, Q.PBLinear)]
    assert len(lins) == 14 and {(m.out_features, m.in_features) for m in lins} == set(LLAMA7B.values())
    assert Q.GEMM_BACKEND == "auto"
    ids = torch.from_numpy((synth.uniform01(2048, 6, 1) * 2048).astype(np.int64)).view(1, -1).to(DEV)
    with torch.no_grad():
        ref = dense(ids, use_cache=False).logits.float()
        out = pb(ids, use_cache=False).logits.float()
    assert all(getattr(m.packed, "_gemm_image", (None, None))[1] is not None for m in lins)     # every linear multiplied from its image
    rel = float((out - ref).abs().max() / ref.abs().max())
    assert rel < 1e-2, rel                                                  # two decoder layers of fp16 arithmetic, other summation order
    agree = float((out.argmax(-1) == ref.argmax(-1)).float().mean())
    assert agree > 0.98, agree
    # decode: fused q/k/v + gate/up launches, one token per forward, captured once and replayed
    n_img, nbytes = H.build_gemm_images_(pb)
    assert n_img == 14 and nbytes > 0
    assert H.fuse_decode_(pb) == 4
    gf = H.GraphedForward(pb, ids[:, :1].clone())
    with torch.no_grad():
        for t in range(8):
            tok = ids[:, 100 + t:101 + t]
            got = gf.replay(tok).float().clone()
            want = dense(tok, use_cache=False).logits.float()
            assert float((got - want).abs().max() / want.abs().max()) < 1e-2, t


# ------------------------------------------------------------------------------------------- config 5
@pytest.mark.parametrize("world,lf", [(2, 0.95), (4, 0.95), (8, 0.95), (2, 0.9), (8, 0.9)])
def test_config5_ksplit_shards_on_one_gpu(world, lf):
    """configs[4], K split (o_proj / down_proj: "row-sharded ... all-reduce"): all P shards of llama-7b down_proj built
    on cuda:0, the HIP kernel run on each with fp32 partial outputs, partials summed in rank order = what the
    all-reduce computes; equals the unsharded HIP result and the oracle.  K = 11008 = 86 x 128 does not divide evenly.
    low_frac 0.9 is the value BASELINE configs[4] names; 0.95 shares its layers with the config 3 tests."""
    layer, W16, mask = cfg3_layer("down_proj", lf)
    N, K = LLAMA7B["down_proj"]
    pts = PP.split_points(K, world, PP.COL_ALIGN)
    assert pts[-1] == K and all(p % 128 == 0 for p in pts[:-1]) and len(set(np.diff(pts))) <= 2
    rh = _cache["down_proj_rtn" if lf == 0.95 else f"down_proj@{lf}_rtn"]
    mods = []
    for rank in range(world):                                       # the P shards, built once (quantizer state handed through)
        shard, (c0, c1) = PP.shard_linear(W16, None, torch.from_numpy(mask), "k", rank, world, high_scale=rh["hscale"],
                                          high_zero=rh["hzero"])
        assert (c0, c1) == (pts[rank], pts[rank + 1])
        mods.append((PP.PBLinearKSplit(shard.to(DEV), (c0, c1)), c0, c1))
    for M in (1, 8):
        x = synth.activations((M, K), 5 + world, 21)
        xt = T(x)
        total = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        for mod, c0, c1 in mods:
            part = mod.local_forward(xt[:, c0:c1].contiguous())
            assert part.dtype == torch.float32 and part.shape == (M, N)
            total += part
        ref = O.dense_linear(x, W16.numpy())
        assert_parity(total, ref, 2e-4)                                    # fp32 sum of fp32 partials
        assert_parity(total.half(), layer(xt).float().cpu().numpy().astype(np.float64), 2e-3)   # vs the unsharded HIP result


@pytest.mark.parametrize("world", [2, 8])
def test_config5_nsplit_shards_on_one_gpu(world):
    """configs[4], N split (q/k/v/gate/up): each rank owns a slice of the output rows; concatenated shard outputs are the
    unsharded output BIT FOR BIT (a record is 16 rows and rows never interact)."""
    layer, W16, mask = cfg3_layer("gate_proj")
    N, K = LLAMA7B["gate_proj"]
    x = T(synth.activations((2, K), 9, 21))
    full = layer(x)
    parts = []
    for rank in range(world):
        rh = _cache["gate_proj_rtn"]
        shard, (r0, r1) = PP.shard_linear(W16, None, torch.from_numpy(mask), "n", rank, world, high_scale=rh["hscale"],
                                          high_zero=rh["hzero"])
        assert r0 % 16 == 0
        parts.append(PP.PBLinearNSplit(shard.to(DEV), (r0, r1), N, gather_output=False)(x))
    assert torch.equal(torch.cat(parts, -1), full)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tp_worker(rank, world, port, collective, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = torch.device(f"cuda:{rank % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if torch.cuda.device_count() >= world else "gloo", rank=rank, world_size=world)
    try:
        N, K = 256, 2048
        W = synth.llm_weight(N, K, seed=41, heavy_tail=True)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        W16 = torch.from_numpy(r["W_fq"]).half()
        x = torch.from_numpy(synth.activations((3, K), 6, 21)).to(dev)
        shard, cols = PP.shard_linear(W16, None, torch.from_numpy(mask), "k", rank, world)
        mod = PP.PBLinearKSplit(shard.to(dev), cols, collective=collective)
        y = mod(x)
        y2 = mod(x)                       # second call: the p2p buffers alternate, results must repeat
        ref = O.dense_linear(x.cpu().numpy(), W16.numpy())
        rel, ratio = O.parity_errors(y.float().cpu().numpy(), ref)
        same = bool(torch.equal(y, y2))
        if mod.comm is not None:
            for _ in range(5):            # more calls than buffer sets
                same = same and bool(torch.equal(mod(x), y))
            # round 4: <= 4 rows take the FUSED pair (GEMV epilogue pushes the partial to every rank + reduce kernel); the unfused
            # pair (GEMV -> y in HBM -> one-shot all-reduce) on the same communicator gives the same bits, in any interleaving
            assert mod.fuse_push
            mod.fuse_push = False
            y_unf = mod(x)
            mod.fuse_push = True
            same = same and bool(torch.equal(y_unf, y)) and bool(torch.equal(mod(x), y))
            x8 = torch.from_numpy(synth.activations((8, K), 16, 21)).to(dev)        # more rows than one GEMV pass: unfused
            r8, q8 = O.parity_errors(mod(x8).float().cpu().numpy(), O.dense_linear(x8.cpu().numpy(), W16.numpy()))
            same = same and r8 < 1e-3 and q8 < 1.0 and bool(torch.equal(mod(x), y))
            # a second K-split layer shares the communicator (one buffer per rank, sized to the largest message) ...
            W2 = synth.llm_weight(128, K, seed=43, heavy_tail=True)
            mask2 = O.ptq_low_mask(W2, 0.9, "magnitude", None, -1)
            W2q = torch.from_numpy(O.ptq_rtn(W2, mask2, 8, -1)["W_fq"]).half()
            shard2, cols2 = PP.shard_linear(W2q, None, torch.from_numpy(mask2), "k", rank, world)
            mod2 = PP.PBLinearKSplit(shard2.to(dev), cols2, collective="p2p")
            same = same and mod2.comm is mod.comm
            # ... and layer + all-reduce are hipGraph-capturable: the call number lives in the buffer, so a REPLAY advances it.
            # Both layers (two dependent all-reduces per replay) captured once, replayed 10 x with fresh inputs, against eager.
            xs = torch.zeros_like(x)
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_), torch.no_grad():
                for _ in range(2):
                    mod2(xs); mod(xs)
            torch.cuda.current_stream().wait_stream(s_)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                ya = mod(xs)
                yb = mod2(xs * 0.5)
            for it in range(10):
                xi = torch.from_numpy(synth.activations((3, K), 60 + it, 21)).to(dev)
                xs.copy_(xi)
                g.replay()
                torch.cuda.synchronize()
                with torch.no_grad():
                    same = same and bool(torch.equal(ya, mod(xi))) and bool(torch.equal(yb, mod2(xi * 0.5)))
            rel2, ratio2 = O.parity_errors(yb.float().cpu().numpy(), O.dense_linear((xs * 0.5).cpu().numpy(), W2q.numpy()))
            same = same and rel2 < 1e-3 and ratio2 < 1.0
            mod.comm.check()
            mod.comm.close()
        out[rank] = (rel, ratio, same)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config5_ksplit_rccl_all_reduce(world):
    """one process per GPU, RCCL all-reduce of the fp32 partials (needs `world` GPUs; the driver's multi-GPU box)"""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(world, _free_port(), "rccl", out), nprocs=world, join=True)
    for rank in range(world):
        rel, ratio, same = out[rank]
        assert rel < 1e-3 and ratio < 1.0 and same


def test_config5_ksplit_p2p_all_reduce_two_processes():
    """libpbl's one-shot peer-to-peer all-reduce between two PROCESSES.  On a one-GPU box both ranks use cuda:0: the
    hipIpc mapping, the slot / flag protocol and the rank-ordered sum are exactly what runs over xGMI; only the link
    differs.  (Handles travel over gloo there, RCCL refuses two ranks on one device.)  Also: two K-split layers on one shared
    communicator, and K-split layer + all-reduce captured in a hipGraph in both processes and replayed 10 times."""
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(2, _free_port(), "p2p", out), nprocs=2, join=True)
    for rank in range(2):
        rel, ratio, same = out[rank]
        assert rel < 1e-3 and ratio < 1.0 and same
    assert out[0][0] == out[1][0]                    # both ranks hold the same bits


def test_config5_ksplit_p2p_four_processes_one_device():
    """the same protocol at four ranks (round 5: slot strides, flag / counter indexing and the agreed push limit at more than two
    peers; eight ranks run as a bench.py plumbing line, profiles/r05_tp8_plumbing.jsonl): four processes share cuda:0, fused push +
    reduce, the unfused all-reduce, two layers on one communicator, graph replay -- every rank holds the same bits"""
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(4, _free_port(), "p2p", out), nprocs=4, join=True)
    for rank in range(4):
        rel, ratio, same = out[rank]
        assert rel < 1e-3 and ratio < 1.0 and same, (rank, rel, ratio, same)
    assert len({out[r][0] for r in range(4)}) == 1


# ------------------------------------------------------------------------------------------- (f1) on-disk format
class _TwoLinears(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.q_proj = torch.nn.Linear(1024, 256, bias=False)
        self.fc = torch.nn.Linear(1024, 128, bias=True)


def test_checkpoint_roundtrip_bit_for_bit(tmp_path):
    """save_pb -> load_pb (utils.py:65-124 save_bnn / load_bnn): the HIP forward of the loaded model equals the forward
    before saving bit for bit, for the GEMV, the matrix-core kernel and the GEMM regime"""
    torch.manual_seed(0)
    model = _TwoLinears()
    side = {}
    for name, lin in (("q_proj", model.q_proj), ("fc", model.fc)):
        W = synth.llm_weight(*lin.weight.shape, seed=len(side) + 60, heavy_tail=True)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        lin.weight.data = torch.from_numpy(r["W_fq"]).half().float()
        side[name] = dict(low_mask=torch.from_numpy(mask), hscale=r["hscale"], hzero=r["hzero"])
    from pb_llm_amd.harness import to_pb_
    model = to_pb_(model.half(), side, skip=()).to(DEV)
    xs = [T(synth.activations((M, 1024), 3 + M, 21)) for M in (1, 16, 70)]
    before = [(model.q_proj(x), model.fc(x)) for x in xs]
    meta = pbio.save_pb(model, str(tmp_path / "ckpt"))
    assert set(meta) == {"q_proj", "fc"} and meta["fc"]["class"] == "PBLinear"
    fresh = pbio.load_pb(_TwoLinears().half(), str(tmp_path / "ckpt")).to(DEV)
    assert isinstance(fresh.q_proj, Q.PBLinear) and fresh.fc.pbl_bias is not None
    for x, (a, b) in zip(xs, before):
        assert torch.equal(fresh.q_proj(x), a) and torch.equal(fresh.fc(x), b)
    # load_state_dict copies INTO the blob buffer: the module must re-derive its launch metadata from the new blob
    W2 = synth.llm_weight(256, 1024, seed=99, heavy_tail=True)
    m2 = O.ptq_low_mask(W2, 0.8, "magnitude", None, -1)               # denser salients: larger max_nch
    r2 = O.ptq_rtn(W2, m2, 8, -1)
    other = Q.PBLinear.from_dense(torch.from_numpy(r2["W_fq"]).half(), None, torch.from_numpy(m2), -1, r2["hscale"], r2["hzero"])
    if other.pbl_blob.numel() == fresh.q_proj.pbl_blob.numel():
        fresh.q_proj.load_state_dict(other.state_dict())
        assert fresh.q_proj.packed.max_nch == other.packed.max_nch
    # a corrupt file must be rejected by the loader, not fed to the kernels
    w = torch.load(str(tmp_path / "ckpt" / "weights.pth"), weights_only=True)
    blob = w["q_proj_blob"].clone()
    blob[80 + 4] ^= 0x40                                                  # rb_info[0].nfull
    w["q_proj_blob"] = blob
    torch.save(w, str(tmp_path / "ckpt" / "weights.pth"))
    with pytest.raises(_lib.PblError):
        pbio.load_pb(_TwoLinears().half(), str(tmp_path / "ckpt"))


# ------------------------------------------------------------------------------------------- (a8) Hessian-mask module
def test_hessian_mask_module_on_gpu(tmp_path, monkeypatch):
    """BinaryXnorExceptOutliersLinearHessian (quant/outlier_quantizer.py:126-143) with a gptq_pb mask file: outlier_mask =
    ~mask, binary_scale stays None (an eval() forward fails like the reference's `sign(W) * None`), the first train()
    forward computes it, and it persists into eval()."""
    monkeypatch.chdir(tmp_path)
    N, K = 96, 1024
    W = synth.llm_weight(N, K, seed=71, heavy_tail=True)
    low = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    pbio.save_low_mask(torch.from_numpy(low), 0.9, "model/layers/0/q_proj")
    m = Q.BinaryXnorExceptOutliersLinearHessian(torch.from_numpy(W).clone(), None, 0.1).to(DEV)
    m.global_name = "model/layers/0/q_proj"
    m.gen_outlier_mask()
    assert m.outlier_mask.is_cuda and torch.equal(m.outlier_mask.cpu(), ~torch.from_numpy(low)) and m.binary_scale is None
    W_hat = O.weight_quant_8bit(W)
    np.testing.assert_array_equal(m.weight.data.cpu().numpy(), W_hat)
    x = synth.normal((3, K), 4, 5, 1.0)
    m.eval()
    with pytest.raises(TypeError):
        m(T(x))
    m.train()
    with torch.no_grad():
        y_train = m(T(x))
    s = O.refresh_binary_scale(W_hat, ~low)
    np.testing.assert_allclose(m.binary_scale.float().cpu().numpy().reshape(-1), np.asarray(s).reshape(-1), rtol=3e-6)
    ref = O.pb_qat_forward(x, W_hat, ~low, np.asarray(s, np.float32).reshape(1, 1))
    assert_parity(y_train, ref, 1e-4)
    m.eval()
    assert_parity(m(T(x)), ref, 2e-5)                     # packed kernels, fp32 module: split-x path


def test_g9_hessian_mask_module_vs_reference_golden(tmp_path, monkeypatch):
    """the same module against outputs of the REFERENCE class (tests/golden/g9, tools/gen_goldens.py G9: the reference's
    BinaryXnorExceptOutliersLinearHessian picks up the mask its own gptq_pb run dumped): mask, 8-bit weights, outlier_nbits,
    binary_scale, the train() forward, the eval() forward through the packed kernels, to_regular_linear; and the magnitude
    fallback when the file is missing."""
    import hashlib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_hessian_mask_module.npz"))
    monkeypatch.chdir(tmp_path)
    N, K = 256, 512
    W = synth.llm_weight(N, K, seed=9, heavy_tail=True).astype(np.float16).astype(np.float32)
    b = synth.normal((N,), 9, 3, 0.1)
    x = synth.normal((3, K), 9, 5, 1.0)
    low = np.unpackbits(g["low_mask"])[:N * K].astype(bool).reshape(N, K)
    pbio.save_low_mask(torch.from_numpy(low), 0.9, "golden/layer")
    m = Q.BinaryXnorExceptOutliersLinearHessian(torch.from_numpy(W).clone(), torch.from_numpy(b), 0.1).to(DEV)
    m.global_name = "golden/layer"
    m.eval()
    m.gen_outlier_mask()
    np.testing.assert_array_equal(np.packbits(m.outlier_mask.cpu().numpy()), g["outlier_mask"])
    assert m.binary_scale is None and bool(g["binary_scale_is_none"])
    np.testing.assert_array_equal(m.weight.data.cpu().numpy(), g["w_hat"])
    assert abs(m.outlier_nbits - float(g["outlier_nbits"])) < 1e-9
    xt = T(x)
    with torch.no_grad():
        m.train()
        y_train = m(xt)
        np.testing.assert_allclose(m.binary_scale.float().cpu().numpy().reshape(-1), g["binary_scale"].reshape(-1), rtol=3e-6)
        assert_parity(y_train, g["y_train"].astype(np.float64), 1e-4)
        m.eval()
        assert_parity(m(xt), g["y_eval"].astype(np.float64), 2e-5)        # packed kernels, fp32 module
        w_sim = m.to_regular_linear().weight.data.float().cpu().numpy()
    assert hashlib.sha256(np.ascontiguousarray(w_sim).tobytes()).hexdigest() == str(g["w_sim_sha"])
    m2 = Q.BinaryXnorExceptOutliersLinearHessian(torch.from_numpy(W).clone(), torch.from_numpy(b), 0.1).to(DEV)
    m2.global_name = "golden/other"                                       # no mask file: magnitude fallback
    m2.eval()
    with torch.no_grad():
        m2.gen_outlier_mask()
        np.testing.assert_array_equal(np.packbits(m2.outlier_mask.cpu().numpy()), g["fallback_mask"])
        assert_parity(m2(xt), g["fallback_y_eval"].astype(np.float64), 2e-5)


def test_g10_reference_save_bnn_directory_runs_on_the_gpu(tmp_path):
    """a checkpoint directory in the reference's save_bnn layout (utils.py:87-94; tests/golden/g10 was written by the
    reference's own save_bnn) loaded by pb_llm_amd.io.load_bnn into the MI355X-backed classes: the forward equals what the
    reference's modules computed after ITS load_bnn round trip"""
    from test_round3_cpu import Net, write_reference_directory
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_save_bnn_directory.npz"))
    write_reference_directory(g, str(tmp_path / "ckpt"))
    net = pbio.load_bnn(Net().to(DEV), str(tmp_path / "ckpt")).eval()
    assert net.fc1.weight.is_cuda and isinstance(net.blk[0], Q.BinaryLinear)
    x = synth.normal((4, 256), 10, 5, 1.0)
    with torch.no_grad():
        assert_parity(net.fc1(T(x)), g["y_fc1_loaded"].astype(np.float64), 2e-5)
        assert_parity(net(T(x)), g["y_loaded"].astype(np.float64), 1e-4)


# ------------------------------------------------------------------------------------------- autograd / cache
def test_packed_forward_input_gradient():
    """the reference's fake-quant nn.Linear is differentiable in x: dx = dy @ W through the packed path, for every regime"""
    N, K = 192, 1024
    W = synth.llm_weight(N, K, seed=81, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    for M in (1, 20, 64):
        x = T(synth.activations((M, K), 8 + M, 21)).requires_grad_(True)
        y = layer(x)
        assert y.requires_grad
        dy = T(synth.activations((M, N), 9 + M, 22))
        y.backward(dy)
        ref = dy.float().cpu().numpy().astype(np.float64) @ W16.float().numpy().astype(np.float64)
        assert_parity(x.grad, ref, 2e-3)
    with torch.no_grad():
        assert not layer(x).requires_grad


def test_packed_cache_follows_the_weight():
    """eval() forwards serve a cached blob; weight.data edits, load_state_dict and dtype casts must invalidate it"""
    W = synth.llm_weight(64, 512, seed=91)
    m = Q.XnorBinaryLinear(torch.from_numpy(W), None).to(DEV).eval()
    x = T(synth.normal((2, 512), 3, 5, 1.0))
    y0 = m(x)
    with torch.no_grad():
        m.weight.mul_(-1.0)                  # (an edit through `.data` bypasses the version counter: call invalidate())
    assert_parity(m(x), -y0.float().cpu().numpy().astype(np.float64), 1e-5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["weight"] = torch.from_numpy(W).to(DEV)
    m.load_state_dict(sd)
    assert torch.equal(m(x), y0)
    q = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W), None, 0.1).to(DEV).eval()
    q.gen_outlier_mask()
    yq = q(x)
    q.half()
    assert q._code_scale.dtype == torch.float32               # the code grid stays fp32 through .half()
    yh = q(x.half())
    assert yh.dtype == torch.float16
    assert_parity(yh, yq.float().cpu().numpy().astype(np.float64), 4e-3)
    assert q._packed.nexc <= 0.01 * q._packed.nnz + 4            # salients still on the code grid, not exceptions


def test_mfma_k_split_equals_unsplit():
    """the K-split matrix-core launch (fp32 partials + fixed-order reduce) against the unsplit one and the oracle"""
    N, K = 512, 4096
    W = synth.llm_weight(N, K, seed=95, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    b = T(synth.normal((N,), 2, 3, 0.1))
    p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"], r["hzero"],
                   (~mask).astype(np.uint8)).to(DEV)
    layer = p.layer_struct(b)
    import ctypes as C
    assert _lib.lib().pbl_mfma_workspace_bytes(C.byref(layer), 32) > 0          # 32 records: the split is in use
    for M in (5, 16, 32):
        x = synth.activations((M, K), 4 + M, 21)
        ref = O.dense_linear(x, r["W_fq"], b.cpu().numpy())
        ys = Q.mfma_forward(p, b, T(x), out_f32=True, split=True)
        yu = Q.mfma_forward(p, b, T(x), out_f32=True, split=False)
        assert_parity(ys, ref, 2e-4)
        assert_parity(yu, ref, 2e-4)
        assert torch.equal(ys, Q.mfma_forward(p, b, T(x), out_f32=True, split=True))    # deterministic


# ------------------------------------------------------------------------------------------- (f4) decode path of the harness
def test_fused_decode_and_graph_replay_bit_for_bit():
    """harness.fuse_decode_: q/k/v and gate/up of every decoder layer become ONE launch each (7 -> 4 per layer); the logits
    of a token-by-token forward equal the unfused model's BIT FOR BIT, also for consecutive tokens whose activations
    reuse the same device address; longer prompts fall through to the members' own kernels; GraphedForward replays the
    captured forward with new token ids and reproduces the eager logits exactly."""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    from pb_llm_amd import harness as H
    from pb_llm_amd.runtime import FusedGemv
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
    plain = H.to_pb_(model, side).to(DEV)
    fused = copy.deepcopy(plain)
    assert H.fuse_decode_(fused) == 4                       # 2 layers x (qkv, gate/up)
    ids = torch.from_numpy((synth.uniform01(64, 7, 1) * 1000).astype(np.int64)).view(1, -1).to(DEV)
    with torch.no_grad():
        for t in range(6):                                  # single tokens, one after the other (M = 1)
            tok = ids[:, t:t + 1]
            assert torch.equal(fused(tok, use_cache=False).logits, plain(tok, use_cache=False).logits), t
        for T_ in (3, 4, 5, 40):                            # <= 4 rows fused, above: the members' own kernels
            assert torch.equal(fused(ids[:, :T_], use_cache=False).logits, plain(ids[:, :T_], use_cache=False).logits), T_
    # the fused launch itself against the oracle, with biases and unequal N
    W1, W2 = synth.llm_weight(96, 512, seed=3), synth.llm_weight(160, 512, seed=4)
    ps, refs, bs = [], [], []
    x = synth.activations((3, 512), 5, 21)
    for i, W in enumerate((W1, W2)):
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        b = synth.normal((W.shape[0],), 6 + i, 3, 0.1)
        ps.append(pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"], r["hzero"],
                             (~mask).astype(np.uint8)))
        bs.append(T(b)); refs.append(O.dense_linear(x, r["W_fq"], b))
    outs = FusedGemv(ps, bs, DEV)(T(x))
    assert outs[0].shape == (3, 96) and outs[1].shape == (3, 160) and outs[0].data_ptr() + 96 * 2 == outs[1].data_ptr()
    for o, ref in zip(outs, refs):
        assert_parity(o, ref)
    # hipGraph: capture one single-token forward, replay with other tokens
    g = H.GraphedForward(fused, ids[:, :1])
    with torch.no_grad():
        for t in (9, 10, 11):
            tok = ids[:, t:t + 1]
            assert torch.equal(g.replay(tok), plain(tok, use_cache=False).logits), t


def test_batched_decode_runs_the_projections_as_one_merged_layer():
    """round 5: with fuse_decode_ a decode BATCH (5 - 64 rows) runs q/k/v (gate/up) as ONE call of their row-wise concatenation
    (packing.concat_rows on the device: byte surgery, validated; harness._FusedGroup.merged) -- one small-batch launch + reduce per
    group instead of one per projection.  The merged layer's blob equals the host-side concatenation byte for byte, its rows are
    the members' rows (against the oracle; another K split than the members' own launches, so tolerance, not bits), a model's
    logits at 16 and 48 rows agree with the unfused model, biases are carried, 2 rows still take the fused GEMV and 300 rows the
    members' own kernels."""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM
    from pb_llm_amd import harness as H
    from pb_llm_amd.packing import concat_rows
    ps, Ws, bs = [], [], []
    for i, N in enumerate((96, 160, 64)):
        W = synth.llm_weight(N, 1024, seed=30 + i)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        lay = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), torch.from_numpy(synth.normal((N,), 7 + i, 3, 0.1)) if i != 1 else None,
                                    torch.from_numpy(mask), -1, r["hscale"], r["hzero"])
        ps.append(lay); Ws.append(r["W_fq"].astype(np.float16).astype(np.float32)); bs.append(lay.pbl_bias)
    host = concat_rows([l.packed for l in ps])
    devl = [l.to(DEV) for l in ps]
    onde = concat_rows([l.packed for l in devl])
    assert torch.equal(onde.blob.cpu(), host.blob) and PackedWeight.from_blob(onde.blob).NRB == 20
    grp = H._FusedGroup(devl)
    x = synth.activations((16, 1024), 3, 21)
    xt = T(x)
    grp.launch(xt, 16, 0, merged=True)
    for o, W, b in zip(grp.outs, Ws, bs):
        assert_parity(o, O.dense_linear(x, W, None if b is None else b.cpu().numpy()))
    assert grp.merged.pbl_bias is not None and grp.merged_launches == 1
    # model level
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
    plain = H.to_pb_(model, side).to(DEV)
    fused = copy.deepcopy(plain)
    assert H.fuse_decode_(fused, merge_batched=True) == 4            # (opt-in since round 6)
    ids = torch.from_numpy((synth.uniform01(320, 7, 1) * 1000).astype(np.int64)).view(1, -1).to(DEV)
    g0 = fused.model.layers[0].self_attn.q_proj._group[0]
    with torch.no_grad():
        for T_ in (16, 48):                                                 # a batch of decode rows: the merged layers
            a, b = fused(ids[:, :T_], use_cache=False).logits.float(), plain(ids[:, :T_], use_cache=False).logits.float()
            assert float((a - b).abs().max() / b.abs().max()) < 5e-3, T_
        assert g0.merged is not None and g0.merged_launches == 2 and g0.served >= 4
        n_merged = g0.merged_launches
        assert torch.equal(fused(ids[:, :2], use_cache=False).logits, plain(ids[:, :2], use_cache=False).logits)      # the fused GEMV launch
        assert torch.equal(fused(ids[:, :300], use_cache=False).logits, plain(ids[:, :300], use_cache=False).logits)  # prefill: the members' own kernels
        assert g0.merged_launches == n_merged
        ab = fused(ids[:, :16], use_cache=False).logits                     # (bf16 hidden states would take the same path; fp16 here)
        assert torch.equal(ab, fused(ids[:, :16], use_cache=False).logits)
    nofuse = copy.deepcopy(plain)
    assert H.fuse_decode_(nofuse) == 4                                  # the default: no merged copies
    with torch.no_grad():
        assert torch.equal(nofuse(ids[:, :16], use_cache=False).logits, plain(ids[:, :16], use_cache=False).logits)   # off: the members' own launches


def test_fused_members_in_any_call_order_and_after_moves():
    """harness._FusedMember: whichever member is called first launches the group, the others are served only for the SAME
    tensor object (held by the group, so a recycled device address cannot match); followers called before the leader,
    across consecutive tokens whose activations reuse one address, give the unfused results bit for bit; a model that
    passes views launches per member (counted); rewriting a member's blob in place rebuilds the launch descriptors."""
    from pb_llm_amd import harness as H
    import torch.nn as nn
    K = 512
    mods, refs = [], []
    for i, N in enumerate((96, 160, 64)):
        W = synth.llm_weight(N, K, seed=30 + i)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        mods.append(Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV))
        refs.append(r["W_fq"])

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj = mods
    holder = Attn()
    plain = [m for m in mods]
    assert H.fuse_decode_(holder) == 1
    fused = [holder.q_proj, holder.k_proj, holder.v_proj]
    grp = fused[0]._group[0]
    for step, order in enumerate([(1, 0, 2), (2, 1, 0), (0, 1, 2), (1, 2, 0)]):
        x = T(synth.activations((1, K), 50 + step, 21))           # a fresh tensor per "token": the allocator may reuse the address
        outs = {i: fused[i](x) for i in order}
        for i in range(3):
            assert torch.equal(outs[i], plain[i](x)), (step, i)
        del x, outs
    assert grp.launches == 4 and grp.served == 8 and grp.solo_launches == 0
    # a different tensor object of the same storage is NOT served from the cache: every member launches
    x = T(synth.activations((2, K), 60, 21))
    before = grp.launches
    for i in (1, 0, 2):
        assert torch.equal(fused[i](x.view(2, K)), plain[i](x))
    assert grp.launches == before + 3 and grp.solo_launches >= 2
    # an in-place edit of x between two member calls invalidates the cached outputs
    ya = fused[0](x)
    x.mul_(2)
    assert torch.equal(fused[1](x), plain[1](x)) and not torch.equal(fused[0](x), ya)
    # load_state_dict into a member rewrites its blob in place: the group rebuilds its descriptors
    W = synth.llm_weight(160, K, seed=99)
    mask = O.ptq_low_mask(W, 0.8, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    other = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])
    if other.pbl_blob.numel() == mods[1].pbl_blob.numel():       # (same size needed for an in-place copy; else skip this leg)
        mods[1].load_state_dict(other.state_dict())
    x2 = T(synth.activations((1, K), 61, 21))
    got = [fused[i](x2) for i in (2, 1, 0)][::-1]
    for i in range(3):
        assert torch.equal(got[i], plain[i](x2))
    assert_parity(got[1], O.dense_linear(x2.cpu().numpy(), mods[1].weight.float().cpu().numpy()))


# ------------------------------------------------------------------------------------------- GEMM regime, fused kernel
@pytest.mark.parametrize("N,K,M,lf,bias", [(256, 512, 33, 0.9, True), (130, 1288, 300, 0.8, False), (4096, 4096, 2048, 0.95, False),
                                           (1000, 11008, 257, 0.95, True)])
def test_fused_gemm_kernel(N, K, M, lf, bias):
    """pbl_gemm_f16: exact fp16 weight tiles rebuilt in LDS from the packed records, v_mfma_f32_16x16x32_f16 against
    x staged in LDS -- against the float64 oracle (sampled rows at the large shape) and against the library GEMM on the
    unpacked layer (same operands, different summation order).  Ragged N / K / M, exceptions, bias."""
    W = synth.llm_weight(N, K, seed=N + K, heavy_tail=True)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    W16[min(5, N - 1), K // 2 + 1] = 0.4321                       # an exception
    W16[N - 1, K - 1] = -0.0625
    b = synth.normal((N,), 2, 3, 0.1) if bias else None
    layer = Q.PBLinear.from_dense(W16, T(b).cpu() if bias else None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    assert layer.packed.flags & _lib.PBL_FLAG_SAL_F16 and layer.packed.nexc >= 1
    x = synth.activations((M, K), N, 21)
    xt = T(x)
    y = Q.fused_gemm_forward(layer.packed, layer.pbl_bias, xt)
    assert y.shape == (M, N) and y.dtype == torch.float16
    rows = np.arange(N) if N * K * M < 4e9 else np.unique(np.concatenate([np.arange(0, N, N // 160), [N - 1, N - 17, 127, 128]]))
    ref = O.dense_linear(x, W16.numpy()[rows], None if b is None else b[rows])
    assert_parity(y[:, torch.from_numpy(rows).to(DEV)], ref)
    old = Q.GEMM_BACKEND, Q.GEMM_SPLIT_K
    try:
        Q.GEMM_BACKEND = "library"
        y_lib = layer(xt)
        Q.GEMM_BACKEND, Q.GEMM_SPLIT_K = "fused", False            # one launch over the image: the round-3 kernel's bits
        assert torch.equal(layer(xt), y)                            # the module routes the GEMM regime to the fused kernel
        Q.GEMM_SPLIT_K = True                                       # the default: few tiles are split along K (other summation order)
        assert_parity(layer(xt), y.float().cpu().numpy().astype(np.float64), 2e-3)
    finally:
        Q.GEMM_BACKEND, Q.GEMM_SPLIT_K = old
    assert_parity(y, y_lib.float().cpu().numpy().astype(np.float64), 2e-3)
    assert torch.equal(y, Q.fused_gemm_forward(layer.packed, layer.pbl_bias, xt))      # deterministic


# ------------------------------------------------------------------------------------------- fp16-checkpoint zeros in the matrix-core tile
def test_mfma_fp16_checkpoint_zero_valued_salients():
    """The matrix-core kernel keeps MINUS the salient weight in its fp16 tile and tells entries from empty positions by "any
    bit set": a salient of value 0 must come out as -0.  Rows with many exact zeros (q == zero point), a row with a non-integer
    zero point and a scale so small that every product underflows to 0 in fp16, and a row of scale 0: the packers code only
    entries whose negated product has a bit set (the rest become exceptions), the blob validates, unpacks bit-exactly and
    every forward path agrees with the oracle."""
    N, K = 32, 1024
    rng = np.random.default_rng(5)
    W = np.where(rng.random((N, K)) < 0.5, 0.25, -0.125).astype(np.float32)
    ss = np.full(N, 0.01, np.float32); sz = np.full(N, 100.0, np.float32)
    sal = rng.random((N, K)) < 0.15
    q = rng.integers(0, 256, (N, K))
    q[:, ::7] = 100                                           # plenty of exact zeros
    ss[3] = 1e-9; sz[3] = 5.3                                 # every product underflows: the weight is +-0
    ss[4] = 0.0                                               # degenerate quantizer
    vals = (ss[:, None] * (q.astype(np.float32) - sz[:, None])).astype(np.float32).astype(np.float16).astype(np.float32)
    W[sal] = vals[sal]
    hi = np.full((N, 1), 0.25, np.float32); lo = np.full((N, 1), -0.125, np.float32)
    p = pack_dense(W, hi, lo, ss, sz, sal.astype(np.uint8), sal_f16=True)
    np.testing.assert_array_equal(p.unpack().numpy(), W)
    assert p.nexc > 0                                          # the entries the tile could not represent
    assert int((W[sal] == 0).sum()) > 300
    pd = p.to(DEV)
    for M in (1, 3, 16, 32):
        x = synth.activations((M, K), 3 + M, 21)
        ref = O.dense_linear(x, W)
        assert_parity(Q.mfma_forward(pd, None, T(x), out_f32=True), ref, 2e-4)
        assert_parity(Q.PBLinear(pd, None)(T(x)), ref)
    # the device packer applies the same rule: byte-identical blob
    pdv = pack_dense_dev_like(W, hi, lo, ss, sz, sal)
    assert torch.equal(pdv.blob.cpu(), p.blob.cpu())


def pack_dense_dev_like(W, hi, lo, ss, sz, sal):
    from pb_llm_amd.packing import pack_dense_dev
    return pack_dense_dev(T(W), T(hi), T(lo), T(ss), T(sz), T(sal.astype(np.uint8)), sal_f16=True)


def test_routing_of_few_tokens_on_wide_layers():
    """pbl_linear_f16_ws routing (tools/bench_route.py): a GEMV pass that would leave one workgroup per CU (K = 11008 with 3-4
    tokens) goes to the matrix-core kernel, which then NEEDS its K-split workspace -- pbl_linear_workspace_bytes tells the
    caller how much; without one, up to 8 tokens fall back to GEMV passes.  All routes agree with the oracle, and the module
    (which asks for the workspace) takes well under the unsplit kernel's time."""
    N, K = 2048, 11008
    W = synth.llm_weight(N, K, seed=9)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    ls = layer.packed.layer_struct(None)
    L = _lib.lib()
    assert L.pbl_linear_workspace_bytes(C.byref(ls), 1) == 0 and L.pbl_linear_workspace_bytes(C.byref(ls), 2) == 0
    need = L.pbl_linear_workspace_bytes(C.byref(ls), 9)
    assert need == L.pbl_mfma_workspace_bytes(C.byref(ls), 9) > 0
    st = torch.cuda.current_stream().cuda_stream
    for M in (3, 4, 7, 9):
        x = synth.activations((M, K), 40 + M, 21)
        ref = O.dense_linear(x, W16.float().numpy())
        assert_parity(layer(T(x)), ref)                                           # module: workspace provided
        y = torch.empty(M, N, dtype=torch.float16, device=DEV)
        _lib.check(L.pbl_linear_f16(C.byref(ls), T(x).data_ptr(), y.data_ptr(), M, 0, st), "linear")   # no workspace
        assert_parity(y, ref)



def test_bf16_activations_single_pass():
    """bf16 -> fp16 is exact inside fp16's range, so bf16 activations take ONE pass (round 1 ran two fp16 terms and spilled to
    the dense path from 17 tokens on): every token count up to 32 stays on the packed kernels, the result equals the oracle on
    the bf16 inputs to fp32-accumulation accuracy before the final bf16 rounding, and values beyond fp16's range saturate
    instead of producing inf / nan."""
    N, K = 512, 2048
    W = synth.llm_weight(N, K, seed=31)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    W16 = torch.from_numpy(r["W_fq"]).half()
    layer = Q.PBLinear.from_dense(W16, None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to(DEV)
    for M in (1, 4, 17, 32, 40):
        xb = T(synth.activations((M, K), 70 + M, 21)).bfloat16()
        ref = O.dense_linear(xb.float().cpu().numpy(), W16.float().numpy())
        y32 = Q.pb_linear_forward(layer.packed, None, xb, out_f32=True)
        assert y32.dtype == torch.float32
        assert_parity(y32, ref, 2e-4 if M <= 32 else 2e-3)          # (40 rows: dense workspace + library GEMM in fp32)
        y = layer(xb)
        assert y.dtype == torch.bfloat16
        np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=1e-2, atol=2e-2)
    xb = torch.zeros(2, K, dtype=torch.bfloat16, device=DEV)
    xb[0, 5] = 1e6; xb[1, 7] = -3e38
    assert torch.isfinite(layer(xb).float()).all()


def test_fused_launch_inline_and_table_descriptors_agree():
    """up to 4 fused members travel in the kernel arguments (pbl_gemv_f16_fused_host), more through the device table
    (pbl_gemv_f16_fused): both equal the members' own launches bit for bit"""
    from pb_llm_amd.runtime import FusedGemv
    K = 1024
    ps, xs = [], synth.activations((2, K), 12, 21)
    for i, N in enumerate((64, 96, 48, 128, 80)):
        W = synth.llm_weight(N, K, seed=40 + i)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        ps.append(pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"], r["hzero"],
                             (~mask).astype(np.uint8)).to(DEV))
    f5, f3 = FusedGemv(ps, None, DEV), FusedGemv(ps[:3], None, DEV)
    assert not f5._inline and f3._inline
    o5, o3 = f5(T(xs), out_f32=True), f3(T(xs), out_f32=True)
    for a, b in zip(o3, o5[:3]):
        assert torch.equal(a, b)
    for p, o in zip(ps, o5):
        assert_parity(o, O.dense_linear(xs, p.unpack().numpy()), 2e-4)


def test_image_only_residency_one_copy_of_the_weights():
    """VERDICT r5 item 5 / missing #6: the fast paths multiplied from a GEMM image that sat ON TOP of the blob.  Round 6:
    PBLinear.release_blob_ makes the image the layer's only device copy -- the blob moves to host memory (still the module's buffer:
    state_dict / load_state_dict work), every row count and activation dtype multiplies from the image (<= 64 rows: small-batch kernel,
    beyond: GEMM kernel) -- and restore_blob_ brings the GEMV back with the same image.  Reference consumers: the perplexity loops
    (gptq_pb/eval_ppl_utils.py:55-64, qat/eval_after_qat.py:11-33), which only call the layers with whole windows."""
    from pb_llm_amd import harness as H

    def make(seed):
        W = synth.llm_weight(512, 1024, seed=seed, heavy_tail=True)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        b = synth.normal((512,), 3, seed, 0.1)
        lay = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), torch.from_numpy(b), torch.from_numpy(mask), -1, r["hscale"], r["hzero"])
        return lay, r["W_fq"].astype(np.float16).astype(np.float32), b

    layer, Wd, b = make(21)
    layer = layer.to(DEV)
    xs = {M: T(synth.activations((M, 1024), 5, M)) for M in (1, 3, 20, 64, 65, 300)}
    before = {M: layer(x) for M, x in xs.items()}
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    blob_bytes = layer.packed.nbytes
    assert layer.release_blob_() == blob_bytes and layer.release_blob_() == 0
    torch.cuda.synchronize()
    assert not layer.pbl_blob.is_cuda and m0 - torch.cuda.memory_allocated() >= 0.95 * blob_bytes       # the blob's device bytes are gone
    assert set(layer.state_dict()) == {"pbl_blob", "pbl_bias"} and layer.state_dict()["pbl_blob"].numel() == blob_bytes
    for M, x in xs.items():
        y = layer(x)
        assert y.dtype == torch.float16 and y.shape == (M, 512)
        assert_parity(y, O.dense_linear(x.cpu().numpy(), Wd, b))
        if M >= 5:
            assert torch.equal(y, before[M]), M                      # the same kernels over the same image as before the release
        assert not (called_ops(lambda: layer(x)) & LIBRARY_GEMM_OPS)
    for M in (1, 40, 300):                                           # bf16 and fp32 activations from the image alone
        xb = xs[300][:M].bfloat16()
        yb = layer(xb)
        assert yb.dtype == torch.bfloat16
        assert O.parity_errors(yb.float().cpu().numpy(), O.dense_linear(xb.float().cpu().numpy(), Wd, b))[0] < 1e-2
        xf = xs[300][:M].float() * 1.0009765625
        yf = layer(xf)
        assert yf.dtype == torch.float32 and O.parity_errors(yf.cpu().numpy(), O.dense_linear(xf.cpu().numpy().astype(np.float64), Wd, b))[0] < 3e-4
    with pytest.raises(_lib.PblError):
        layer(xs[3].clone().requires_grad_(True))                    # the input gradient needs the blob
    # new weights through load_state_dict: the host buffer is rewritten, the image follows on the next call
    other, Wd2, b2 = make(22)
    layer.load_state_dict(other.state_dict())
    assert_parity(layer(xs[300]), O.dense_linear(xs[300].cpu().numpy(), Wd2, b2))
    assert_parity(layer(xs[1]), O.dense_linear(xs[1].cpu().numpy(), Wd2, b2))
    layer.restore_blob_()
    assert layer.pbl_blob.is_cuda and layer._image_only is None
    ref1 = other.to(DEV)
    assert torch.equal(layer(xs[1]), ref1(xs[1])) and torch.equal(layer(xs[300]), ref1(xs[300]))       # the GEMV again; prefill bit for bit
    # model level: build_gemm_images_(release_blobs=True) on a small HF llama; a fused model refuses
    from transformers import LlamaConfig, LlamaForCausalLM
    import copy
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8,
                      num_key_value_heads=8, vocab_size=1000, max_position_embeddings=512)
    model = LlamaForCausalLM(cfg).half().eval()

    def producer(name, W):
        Wn = W.float().numpy()
        mask = O.ptq_low_mask(Wn, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(Wn, mask, 8, -1)
        return dict(W_fq=torch.from_numpy(r["W_fq"]), low_mask=torch.from_numpy(mask), hscale=r["hscale"], hzero=r["hzero"])

    side = H.quantize_dense_(model, producer)
    plain = H.to_pb_(model, side).to(DEV)
    lean = copy.deepcopy(plain)
    packed_bytes = sum(m.packed.nbytes for m in lean.modules() if isinstance(m, Q.PBLinear))
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated()
    n, img_bytes = H.build_gemm_images_(lean, release_blobs=True)
    torch.cuda.synchronize()
    assert n == 14 and torch.cuda.memory_allocated() - m1 <= img_bytes - 0.95 * packed_bytes + (1 << 20)      # images in, blobs out
    ids = torch.from_numpy((synth.uniform01(320, 7, 1) * 1000).astype(np.int64)).view(1, -1).to(DEV)
    with torch.no_grad():
        for T_ in (300, 16, 1):
            a, c = lean(ids[:, :T_], use_cache=False).logits.float(), plain(ids[:, :T_], use_cache=False).logits.float()
            assert float((a - c).abs().max() / c.abs().max()) < 5e-3, T_
    assert H.fuse_decode_(lean) == 0                                    # image-only projections are not fused (the fused launches read blobs)
    fused = copy.deepcopy(plain)
    H.fuse_decode_(fused)
    with pytest.raises(RuntimeError):
        H.release_blobs_(fused)
    assert H.restore_blobs_(lean) == 14
    with torch.no_grad():
        assert torch.equal(lean(ids[:, :1], use_cache=False).logits, plain(ids[:, :1], use_cache=False).logits)
