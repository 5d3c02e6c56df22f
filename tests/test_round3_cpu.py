"""CPU tests added in round 3 (no GPU, no kernels): the reference's own checkpoint directory layout (utils.py:87-124) read and
written by pb_llm_amd.io, against a directory the REFERENCE wrote (tests/golden/g10, tools/gen_goldens.py G10); the C ABI
symbols of this round; the GEMM-regime workspace sizing on the host."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from pb_llm_amd import _lib, io as pbio, quant as Q, synth
from pb_llm_amd.packing import PackedWeight
from conftest import golden


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(256, 128, bias=True)
        self.blk = nn.Sequential(nn.Linear(128, 64, bias=False))

    def forward(self, x):
        return self.blk(self.fc1(x))


def write_reference_directory(g, path):
    """the files of the reference's save_bnn, rebuilt from the fixture's arrays"""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "meta.json"), "w") as f:
        f.write(str(g["meta_json"]))
    weights = {}
    for k in json.loads(str(g["keys"])):
        weights[k] = None if bool(g["none__" + k]) else torch.from_numpy(np.array(g["w__" + k]))
    torch.save(weights, os.path.join(path, "weights.pth"))
    return weights


def test_load_bnn_reads_a_directory_written_by_the_reference(tmp_path):
    g = golden("g10_save_bnn_directory")
    weights = write_reference_directory(g, str(tmp_path / "ckpt"))
    net = pbio.load_bnn(Net(), str(tmp_path / "ckpt"))
    assert isinstance(net.fc1, Q.XnorBinaryLinear) and isinstance(net.blk[0], Q.BinaryLinear)
    assert isinstance(net.fc1, Q.BinaryInterface) and net.fc1.global_name == "fc1" and net.blk[0].global_name == "blk/0"
    # the modules hold the stored fp16 weights (as fp32 Parameters, like the reference's constructors)
    assert net.fc1.weight.dtype == torch.float32 and torch.equal(net.fc1.weight.data, weights["fc1_weight"].float())
    assert torch.equal(net.blk[0].weight.data, weights["blk.0_weight"].float()) and net.blk[0].bias is None
    assert torch.equal(net.fc1.bias.data, weights["fc1_bias"].float())
    # and write the same layout back: same keys, same tensors, same meta
    meta = pbio.save_bnn(net, str(tmp_path / "again"))
    assert meta == json.loads(str(g["meta_json"]))
    again = torch.load(str(tmp_path / "again" / "weights.pth"), weights_only=False)
    assert sorted(again) == sorted(weights)
    for k, v in weights.items():
        if v is None:
            assert again[k] is None
        else:
            a = again[k].data if isinstance(again[k], torch.Tensor) else again[k]
            assert torch.equal(a.float(), v.float()), k
    with pytest.raises(ValueError):
        bad = Net(); bad.fc1 = nn.Linear(100, 128)
        pbio.load_bnn(bad, str(tmp_path / "ckpt"))


def test_round3_c_abi_symbols_and_gemm_workspace_sizing():
    import __graft_entry__ as ge
    ge.build()
    L = _lib.lib()
    for sym in ("pbl_gemm_f16_ex", "pbl_gemm_f16_ws", "pbl_gemm_workspace_bytes", "pbl_p2p_allreduce_f32_dev", "pbl_p2p_buffer_bytes_world"):
        assert hasattr(L, sym)
    lay = _lib.PblLayer(blob=None, bias=None, N=4096, K=4096, P=8, G=1, NRB=256, flags=0xE, max_nch=500, max_nexc=3)
    assert L.pbl_gemm_workspace_bytes(C.byref(lay), 256) == 0                 # one token tile: decode inside the kernel
    nb = L.pbl_gemm_workspace_bytes(C.byref(lay), 257)
    cap = (16 * 500 + 3 + 3) & ~3
    assert nb == ((256 * 36 * 4 + 15) & ~15) + 256 * cap * 4                   # entry ranges (33 -> 36 per record) + entry words
    lay.K, lay.P = 16512, 33
    assert L.pbl_gemm_workspace_bytes(C.byref(lay), 2048) == 0                # 129 half slabs: too wide for the range registers
    # slots strided by the world size: a 2-rank communicator needs an eighth of the 16-rank bound
    assert L.pbl_p2p_buffer_bytes_world(4096, 2) < L.pbl_p2p_buffer_bytes(4096) // 4
    assert L.pbl_p2p_buffer_bytes_world(4096, 17) == 0
    assert _lib.native_linear() is not None                                   # the native dispatcher is built and registered


def test_gemm_list_halves_of_the_c_abi_without_a_gpu():
    """pbl_gemm_list_bytes / pbl_gemm_prepare / pbl_gemm_f16_prepared: sizing is a pure function of the layer (never of M), the
    argument checks answer before any launch"""
    import __graft_entry__ as ge
    ge.build()
    L = _lib.lib()
    lay = _lib.PblLayer(blob=None, bias=None, N=4096, K=4096, P=8, G=1, NRB=256, flags=0xE, max_nch=500, max_nexc=3)
    nb = L.pbl_gemm_list_bytes(C.byref(lay))
    assert nb == L.pbl_gemm_workspace_bytes(C.byref(lay), 2048) == L.pbl_gemm_workspace_bytes(C.byref(lay), 257) > 0
    lay.K, lay.P = 16512, 33
    assert L.pbl_gemm_list_bytes(C.byref(lay)) == 0
    lay.K, lay.P = 4096, 8
    assert L.pbl_gemm_prepare(C.byref(lay), None, 0, None) == _lib.PBL_ERR_INVALID_ARG                    # no blob
    assert L.pbl_gemm_f16_prepared(C.byref(lay), None, None, 64, 0, None, 0, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_gemm_list_bytes(None) == 0


def test_decode_regime_routing_of_the_native_dispatcher():
    """pb_linear_forward hands <= 32 fp16 rows to the native operator only where the C router has a kernel for them: layers
    the matrix-core kernel refuses (group size not a power of two >= 128, K % 8) must keep the Python path from 12 rows on,
    which sends them to the dense backend"""
    def pw(K, G, flags=0xE):
        return PackedWeight(torch.zeros(16, dtype=torch.uint8), 64, K, (K + 511) // 512, G, 4, flags, 8, 0, 0, 0)
    assert Q._mfma_ok(pw(4096, 1))
    assert Q._mfma_ok(pw(4096, 32)) and Q._mfma_ok(pw(4096, 16))          # groups of 128 / 256 columns
    assert not Q._mfma_ok(pw(1152, 3))                                      # groups of 384 columns
    assert not Q._mfma_ok(pw(4096, 64))                                     # groups of 64 columns
    assert not Q._mfma_ok(pw(4100, 1))                                      # K % 8
    assert not Q._mfma_ok(pw(4096, 1, flags=0x6))                           # no slab index (format version 1 blob)
    assert Q.GEMM_THRESHOLD == 12 and Q.MFMA_MAX == 32


def test_asm_issued_loads_of_the_gemm_producers_are_never_touched_in_flight():
    """csrc/pbl_gemm_big.hip issues the LIST producers' stage requests from inline asm and waits for them with a counted vmcnt
    (hipcc's own bookkeeping degenerates to vmcnt(0) in that loop).  Nothing but register discipline keeps that correct: no
    instruction may read or write a register between the load that will fill it and the wait that covers it.
    tools/audit_asm_loads.py compiles the file and walks the generated ISA (prologue + two trips around the stage loop)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "audit_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0
