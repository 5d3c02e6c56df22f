"""CPU tests added in round 4 (no GPU): the reference-layout checkpoint directory round-trips for the packed layer class,
per-class constructor keywords in load_bnn, the restricted unpickler."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import pb_oracle as O
from pb_llm_amd import io as pbio, quant as Q, synth


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(256, 128, bias=True)
        self.blk = nn.Sequential(nn.Linear(128, 64, bias=False))


def _pb_layer(N, K, seed, bias):
    W = synth.llm_weight(N, K, seed=seed)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    b = torch.from_numpy(synth.llm_weight(1, N, seed=seed + 1)[0]).float() if bias else None
    return Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), b, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])


def test_save_bnn_load_bnn_round_trip_for_packed_layers(tmp_path):
    """save_bnn writes class "PBLinear" + the dense fp16 weight (get_save_weight_dict, quant/quantizer.py:70-72); load_bnn packs
    it again (exact for any weight) -- round 3's loader called PBLinear(weight, bias) and failed"""
    net = Net()
    net.fc1 = _pb_layer(128, 256, 3, True)
    net.blk[0] = _pb_layer(64, 128, 5, False)
    meta = pbio.save_bnn(net, str(tmp_path / "ckpt"))
    assert meta == {"fc1": "PBLinear", "blk.0": "PBLinear"}
    back = pbio.load_bnn(Net(), str(tmp_path / "ckpt"))
    for a, b in ((net.fc1, back.fc1), (net.blk[0], back.blk[0])):
        assert isinstance(b, Q.PBLinear)
        assert torch.equal(a.weight, b.weight)                 # the dense simulated weights agree bit for bit
        assert (a.bias is None) == (b.bias is None)
        if a.bias is not None:
            assert torch.equal(a.bias, b.bias)
    assert back.fc1.global_name == "fc1" and back.blk[0].global_name == "blk/0"


def test_load_bnn_hands_every_class_only_its_own_keywords(tmp_path):
    """a directory that mixes classes: outlier_fraction reaches BinaryXnorExceptOutliersLinear only (BinaryLinear's constructor
    does not take it and used to raise TypeError)"""
    w1 = torch.from_numpy(synth.llm_weight(128, 256, seed=7)).half()
    w2 = torch.from_numpy(synth.llm_weight(64, 128, seed=8)).half()
    os.makedirs(tmp_path / "mix")
    with open(tmp_path / "mix" / "meta.json", "w") as f:
        json.dump({"fc1": "BinaryXnorExceptOutliersLinear", "blk.0": "BinaryLinear"}, f)
    torch.save({"fc1_weight": w1, "fc1_bias": nn.Parameter(torch.zeros(128)), "blk.0_weight": w2, "blk.0_bias": None},
               str(tmp_path / "mix" / "weights.pth"))
    net = pbio.load_bnn(Net(), str(tmp_path / "mix"), outlier_fraction=0.2)
    assert isinstance(net.fc1, Q.BinaryXnorExceptOutliersLinear) and net.fc1.outlier_fraction == 0.2
    assert isinstance(net.blk[0], Q.BinaryLinear)
    net = pbio.load_bnn(Net(), str(tmp_path / "mix"), class_kwargs={"BinaryXnorExceptOutliersLinear": {"outlier_fraction": 0.1}})
    assert net.fc1.outlier_fraction == 0.1


class _Evil:
    def __reduce__(self):
        return (os.system, ("true",))


def test_load_bnn_refuses_a_pickle_payload_unless_asked(tmp_path):
    os.makedirs(tmp_path / "bad")
    with open(tmp_path / "bad" / "meta.json", "w") as f:
        json.dump({"fc1": "BinaryLinear"}, f)
    torch.save({"fc1_weight": torch.zeros(128, 256).half(), "fc1_bias": None, "x": _Evil()}, str(tmp_path / "bad" / "weights.pth"))
    with pytest.raises(ValueError, match="restricted unpickler"):
        pbio.load_bnn(Net(), str(tmp_path / "bad"))


@pytest.mark.parametrize("gs", [-1, 128])
def test_from_quantizers_vectorised_host_path_against_the_reference_goldens(gs):
    """PBLinear.from_quantizers composes q_high * ~mask + q_low * mask (gptq_pb/gptq.py:119-127) from the quantizer state with
    whole-matrix tensor ops (round 3 looped over the column groups on the host): RTN goldens G5, with and without groups"""
    from conftest import golden
    from test_oracle_golden import g5_inputs, g5_name
    W16, _, _, _ = g5_inputs()
    g = golden(g5_name("magnitude", gs, True, 0.9))
    mask = np.unpackbits(g["mask"])[:768 * 768].astype(bool).reshape(768, 768)
    a = Q.PBLinear.from_quantizers(torch.from_numpy(W16), torch.from_numpy(mask), g["mean"], g["scale"], g["hscale"], g["hzero"],
                                   groupsize=gs, dtype=torch.float16)
    assert a.packed.G == (1 if gs == -1 else 6)
    assert np.count_nonzero(a.weight.numpy() != g["W_fq"]) <= 8


def test_gemm_image_sizing_of_the_c_abi_without_a_gpu():
    """pbl_gemm_image_stats_bytes / _bytes / small-image workspace: pure functions of the layer and the two geometry words; the
    argument checks answer before any launch"""
    import ctypes as C
    import __graft_entry__ as ge
    from pb_llm_amd import _lib
    ge.build()
    L = _lib.lib()
    lay = _lib.PblLayer(blob=None, bias=None, N=4096, K=4096, P=8, G=1, NRB=256, flags=0xE, max_nch=500, max_nexc=3)
    sb = L.pbl_gemm_image_stats_bytes(C.byref(lay))
    assert sb == 16 + 256 * 4 + 256 * 128 * 4                                  # geometry words, record starts, slot tables
    geom = (C.c_uint32 * 2)(8192, 1)                                           # 32 one-KiB slots per record
    nb = L.pbl_gemm_image_bytes(C.byref(lay), geom)
    rtab_off = (64 + 256 * 4 + 255) & ~255
    assert nb == rtab_off + 256 * 512 + 8192 * 256 + 256 * 64                  # header + starts, tables, slots, level rows
    for bad in ((8192, 0), (8192, 6), (8192, 0xFFFFFFFF)):                     # no vectors / more than five / "a slot does not fit"
        assert L.pbl_gemm_image_bytes(C.byref(lay), (C.c_uint32 * 2)(*bad)) == 0
    assert L.pbl_gemm_image_bytes(C.byref(lay), None) == 0
    lay.K, lay.P = 16512, 33                                                   # 129 half slabs: no image
    assert L.pbl_gemm_image_stats_bytes(C.byref(lay)) == 0 and L.pbl_gemm_image_bytes(C.byref(lay), geom) == 0
    lay.K, lay.P = 4096, 8
    assert L.pbl_gemm_image_stats(C.byref(lay), None, None) == _lib.PBL_ERR_INVALID_ARG                       # no blob, no buffer
    assert L.pbl_gemm_image_build(C.byref(lay), geom, None, None, 0, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_gemm_f16_image(C.byref(lay), None, None, 64, 0, None, 0, geom, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_gemm_small_image_ws(C.byref(lay), None, None, 8, 0, None, 0, geom, None, 0, None) == _lib.PBL_ERR_INVALID_ARG
    # the small-batch kernel's K split: only for <= 64 rows; the workspace is KS x M x N floats
    assert L.pbl_gemm_small_image_workspace_bytes(C.byref(lay), 65) == 0 and L.pbl_gemm_small_image_workspace_bytes(C.byref(lay), 0) == 0
    w8, w32, w64 = (L.pbl_gemm_small_image_workspace_bytes(C.byref(lay), m) for m in (8, 32, 64))
    assert w8 > 0 and w32 == 4 * w8 and w64 == 8 * w8 and w8 % (8 * 4096 * 4) == 0 and 2 <= w8 // (8 * 4096 * 4) <= 8         # KS <= NH / 4 = 8


def test_which_calls_get_the_gemm_image_without_a_gpu(monkeypatch):
    """pb_linear_forward's only decision left in Python (quant._route_image): which image, if any, the native operator multiplies
    from, and whether the small-batch kernel may use it.  Checked with the operator, the image build and the device query replaced
    by stubs.  Round 5 policy: backend "auto" is ALWAYS the hand-written kernel -- every fp16-exact layer gets its image in the
    GEMM regime, whatever the shape and the activation dtype; "tuned" keeps round 4's fills-the-chip rule; 5 - 64 kernel rows take
    the small-batch kernel by SMALL_BATCH_IMAGE; never fp32-grid layers, the library backend, or fewer than 5 rows"""
    from pb_llm_amd import _lib
    from pb_llm_amd.packing import PackedWeight

    calls = []

    class FakeTensor:                      # stands in for x: only what the routing reads
        def __init__(self, M, K, dtype):
            self.shape, self.dtype, self.is_cuda, self.device = (M, K), dtype, True, torch.device("cpu")

        def numel(self):
            return self.shape[0] * self.shape[1]

    def fake_native(blob, bias, x, N, K, P, G, NRB, flags, max_nch, max_nexc, out_f32, dense_f16, img, geom, backend, small_ok, split_k,
                    x_fragments=False, xfrag=None):
        calls.append((x.shape[0], img is not None, small_ok, backend))
        return None

    class FakeImage:
        data, geom_list, ready = object(), [8, 1], None

    built = []
    monkeypatch.setattr(_lib, "native_linear", lambda: fake_native)
    monkeypatch.setattr(Q, "_wait_image", lambda stream, image: None)
    monkeypatch.setattr(Q, "GEMM_X_FRAGMENTS", False)       # (the fragment-major copy of x needs a real tensor: routing only here)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: None)
    monkeypatch.setattr(Q, "gemm_image", lambda packed: built.append(1) or FakeImage())
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setitem(Q._CU_COUNT, 0, 256)
    monkeypatch.setattr(torch.Tensor, "_version", 0, raising=False)

    def pw(N=4096, K=4096):
        blob = torch.zeros(16, dtype=torch.uint8)
        return PackedWeight(blob, N, K, (K + 511) // 512, 1, N // 16, 0xE, 8, 0, 0, 0)

    def route(M, dtype=torch.float16, dense_dtype=None, packed=None):
        packed = packed or pw()
        calls.clear()
        Q.pb_linear_forward(packed, None, FakeTensor(M, packed.K, dtype), dense_dtype=dense_dtype)
        return calls[-1][1], calls[-1][2]          # (image handed over, the small-batch kernel may use it)

    def gets_image(*a, **k):
        return route(*a, **k)[0]

    assert Q.GEMM_BACKEND == "auto" and Q.SMALL_BATCH_IMAGE == "1"              # the shipped defaults (unless the environment says otherwise)
    monkeypatch.setattr(Q, "GEMM_BACKEND", "auto"); monkeypatch.setattr(Q, "GEMM_KEEP_IMAGE", True)
    monkeypatch.setattr(Q, "SMALL_BATCH_IMAGE", "auto")
    assert not gets_image(16) and not built                                    # small batch "auto": no image yet, none is built for it
    assert route(48) == (True, False) and len(built) == 1                      # 33 - 64 rows are GEMM regime: the GEMM kernel over a fresh image
    p = pw()
    assert gets_image(2048, packed=p) and len(built) == 2                      # prefill builds and keeps it ...
    assert route(16, packed=p) == (True, True) and route(5, packed=p) == (True, True) and route(64, packed=p) == (True, True) and len(built) == 2   # ... and small batches use it
    assert not gets_image(4, packed=p)                                         # GEMV passes below 5 rows
    assert route(300, packed=p) == (True, False)                               # 2 x 32 tiles of 128 x 256: still hand-written under "auto"
    assert not gets_image(16, dense_dtype=torch.float32, packed=p) and not gets_image(2048, dense_dtype=torch.float32, packed=p)   # an fp32-grid layer: the image holds fp16 weights
    assert route(48, dtype=torch.bfloat16, packed=p) == (True, True) and route(2048, dtype=torch.bfloat16, packed=p) == (True, False)   # bf16 x: the same kernels (round 5)
    assert gets_image(2048, dtype=torch.float32, packed=p)                     # fp32 x: two fp16 terms through the GEMM kernel
    assert gets_image(8, dtype=torch.float32, packed=p) and not gets_image(2, dtype=torch.float32, packed=p)   # fp32 x = 2 M fp16 rows
    assert gets_image(2048, packed=pw(11008, 4096)) and gets_image(2048, packed=pw(5120, 5120)) and gets_image(2048, packed=pw(13824, 5120)) \
        and gets_image(2048, packed=pw(5120, 13824))                           # "auto" never leaves the hand-written kernel: gate / up and every llama-13b shape
    monkeypatch.setattr(Q, "SMALL_BATCH_IMAGE", "1")
    n0 = len(built)
    assert route(16) == (True, True) and len(built) == n0 + 1                  # "1": built on the first small-batch call
    monkeypatch.setattr(Q, "SMALL_BATCH_IMAGE", "0")
    assert not gets_image(16, packed=p) and route(48, packed=p) == (True, False) and gets_image(2048, packed=p)   # "0": the small-batch kernel never runs over the image
    monkeypatch.setattr(Q, "GEMM_BACKEND", "library")
    assert not gets_image(2048, packed=p) and not gets_image(48, packed=p)
    monkeypatch.setattr(Q, "GEMM_BACKEND", "fused"); monkeypatch.setattr(Q, "SMALL_BATCH_IMAGE", "auto")
    assert gets_image(300, packed=p) and gets_image(48, packed=p)
    assert gets_image(2048, packed=pw(11008, 4096))                            # "fused" = "auto"
    monkeypatch.setattr(Q, "GEMM_BACKEND", "tuned")                            # round 4's default: the image kernel only where its tiles fill the chip
    assert gets_image(2048, packed=p) and not gets_image(300, packed=p) and not gets_image(2048, packed=pw(11008, 4096))
    assert calls[-1][3] == "tuned"
