"""CPU tests for round 2: format version 2 (slab index), structural blob validation
(corrupt-blob fuzz), host-side module logic that needs no GPU, bench.py's launcher checks."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import pb_format_ref as F
from oracle import pb_oracle as O
from pb_llm_amd import _lib, synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import PackedWeight, pack_dense

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _layer(N=80, K=1536, lf=0.9, seed=5, f16=False):
    W = synth.llm_weight(N, K, seed=seed, heavy_tail=True)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    Wd = r["W_fq"].astype(np.float16).astype(np.float32) if f16 else r["W_fq"]
    hi, lo = r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0]
    if f16:
        hi, lo = hi.astype(np.float16).astype(np.float32), lo.astype(np.float16).astype(np.float32)
    return Wd, pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=f16)


def test_slab_index_matches_independent_decoder():
    """the decoder re-derives every slab entry from the chunk lists (asserts inside decode); dense, sparse and ragged K"""
    for N, K, lf in ((80, 1536, 0.9), (33, 520, 0.8), (16, 4096, 0.995), (40, 264, 0.5)):
        Wd, p = _layer(N, K, lf, seed=N + K)
        assert p.flags & _lib.PBL_FLAG_SLABS
        np.testing.assert_array_equal(F.decode(p.blob.numpy()), Wd)


def test_slab_index_chunk_spanning_several_slabs():
    """a row with 3 salient entries 700 columns apart... cannot share a chunk (gap > 127); entries 100 apart can: one
    16-entry chunk then covers 1500 columns = 6 slabs, each of which must list it (fback), and only it"""
    N, K = 16, 2048
    W = np.full((N, K), 0.5, np.float32)
    W[:, ::2] = -0.5
    cols = 20 + 100 * np.arange(16)
    W[3, cols] = 0.25 * np.arange(3, 19)            # codes 3..18 on a grid of 0.25 (none equals a level)
    hi, lo = np.full(N, 0.5, np.float32), np.full(N, -0.5, np.float32)
    p = pack_dense(W, hi, lo, np.full(N, 0.25, np.float32), np.zeros(N, np.float32))
    assert p.nnz == 16 and p.nexc == 0
    np.testing.assert_array_equal(F.decode(p.blob.numpy()), W)       # includes the slab-index cross check
    np.testing.assert_array_equal(p.unpack().numpy(), W)


def test_zero_valued_salients_stay_coded():
    """a salient whose value is 0 (code == zero point; the QAT layer's sign(0) = 0 entries) is an ordinary code entry in
    both modes -- the matrix-core kernel stores it as -0 in its tile, the format needs no special case"""
    N, K = 16, 512
    W = np.full((N, K), 0.5, np.float32)
    W[:, 1::2] = -0.5
    sal = np.zeros((N, K), np.uint8)
    W[2, 10], W[2, 11], W[5, 300] = 0.0, 0.75, 0.0
    sal[2, 10] = sal[2, 11] = sal[5, 300] = 1
    hi, lo = np.full(N, 0.5, np.float32), np.full(N, -0.5, np.float32)
    ss, sz = np.full(N, 0.25, np.float32), np.full(N, 4.0, np.float32)          # value = 0.25 (q - 4): q = 4 -> 0
    for f16 in (True, False):
        p = pack_dense(W, hi, lo, ss, sz, sal, sal_f16=f16)
        assert (p.nnz, p.nexc) == (3, 0)
        np.testing.assert_array_equal(F.decode(p.blob.numpy()), W)


def test_corrupt_blobs_are_rejected():
    """pbl_blob_describe walks the whole structure: every single-byte corruption of the tables either leaves a
    structurally valid blob (payload bytes: codes, sign bits, values) or is rejected with BAD_BLOB -- and a blob it
    accepts can be unpacked without touching memory outside it"""
    Wd, p = _layer(48, 1024, 0.9, seed=9, f16=True)
    good = p.blob.numpy().copy()
    L = _lib.lib()
    layer = _lib.PblLayer()
    assert L.pbl_blob_describe(good.ctypes.data, good.size, C.byref(layer)) == 0
    h = F.read_header(good)
    rng = np.random.default_rng(0)
    # 1. header and record-info table: every byte matters
    rejected = 0
    for off in list(range(8, 80 - 12)) + list(range(80, 80 + 16 * (h["NRB"] + 1))):
        bad = good.copy()
        bad[off] ^= 1 << int(rng.integers(8))
        if L.pbl_blob_describe(bad.ctypes.data, bad.size, C.byref(layer)) != 0:
            rejected += 1
    assert rejected >= 0.8 * (60 + 16 * (h["NRB"] + 1))       # (header maxima may grow: they are upper bounds)
    # 2. random corruption anywhere: accepted blobs must unpack in bounds (run under the allocator's guard: a buffer
    #    exactly the blob's size; numpy would not catch an overrun, so the bound is re-checked through the decoder too)
    out = np.empty((h["N"], h["K"]), np.float32)
    accepted = 0
    for _ in range(400):
        bad = good.copy()
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(bad.size))] ^= 1 << int(rng.integers(8))
        if L.pbl_blob_describe(bad.ctypes.data, bad.size, C.byref(layer)) == 0:
            accepted += 1
            assert L.pbl_unpack_dense_f32(bad.ctypes.data, bad.size, out.ctypes.data) == 0
    assert accepted > 0
    # 3. truncation and trailing garbage
    assert L.pbl_blob_describe(good.ctypes.data, good.size - 128, C.byref(layer)) == _lib.PBL_ERR_BAD_BLOB
    longer = np.concatenate([good, np.zeros(128, np.uint8)])
    assert L.pbl_blob_describe(longer.ctypes.data, longer.size, C.byref(layer)) == _lib.PBL_ERR_BAD_BLOB
    # 4. targeted: a chunk whose columns run past K, a slab entry pointing past the row's chunks, a bad exception row
    rb = good[80:80 + 16].view(np.uint32)
    rec = int(rb[0]) * 16
    nfull, ntail, nexc, off_sal = good[rec:rec + 16].view(np.uint32)
    nch = int(nfull + ntail)
    a128 = lambda v: (v + 127) & ~127       # noqa: E731
    bad = good.copy()
    bad[rec + off_sal: rec + off_sal + 2].view(np.uint16)[0] = 1020          # col0 of chunk 0: its 16 entries overrun K = 1024
    assert L.pbl_blob_describe(bad.ctypes.data, bad.size, C.byref(layer)) == _lib.PBL_ERR_BAD_BLOB
    bad = good.copy()
    bad[rec + off_sal + a128(2 * nch) + 1] = 255                              # a delta of 255: odd (steps are stored doubled)
    assert L.pbl_blob_describe(bad.ctypes.data, bad.size, C.byref(layer)) == _lib.PBL_ERR_BAD_BLOB
    with pytest.raises(_lib.PblError):
        PackedWeight.from_blob(torch.from_numpy(bad))


def test_status_codes_exposed():
    assert _lib.PBL_ERR_BAD_BLOB == -2


def test_train_mode_never_packs_and_cache_key_tracks_weight():
    """train(): no packing whatever the grad mode (the first pass of reentrant checkpointing runs under no_grad);
    the eval() cache is keyed on the weight's storage / version / dtype"""
    W = torch.from_numpy(synth.llm_weight(32, 256, seed=3))
    m = Q.XnorBinaryLinear(W, None)
    k0 = m._cache_key()
    with torch.no_grad():
        m.weight.mul_(2.0)
    k1 = m._cache_key()
    assert k0 != k1
    m.weight.data = m.weight.data.clone()
    assert m._cache_key() != k1
    m.train()
    assert m._is_training_step(torch.zeros(1, 256))
    with torch.no_grad():
        assert m._is_training_step(torch.zeros(1, 256))
    m.eval()
    assert not m._is_training_step(torch.zeros(1, 256))
    q = Q.BinaryXnorExceptOutliersLinear(W.clone(), None, 0.1)
    q.gen_outlier_mask()
    ka = q._cache_key()
    q.binary_scale = q.binary_scale.clone()
    assert q._cache_key() != ka
    q.half()
    assert q._code_scale.dtype == torch.float32 and q._code_zp.dtype == torch.float32 and q.weight.dtype == torch.float16


def test_bench_refuses_a_world_size_mismatch():
    """bench.py --gpus N started as one plain process must not print a 1-rank line labelled N: with WORLD_SIZE set to
    something else it exits non-zero before touching the GPU"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_forward_routing_decisions_without_a_gpu():
    """pbl_linear_workspace_bytes is pure host logic (the routing of pbl_linear_f16_ws): > 0 means "this call goes to the
    matrix-core kernel, bring the K-split scratch".  Pins the decisions measured in tools/bench_route.py."""
    L = _lib.lib()

    def layer(N, K, sal_frac, G=1, flags=_lib.PBL_FLAG_SLABS | 0x4):          # 0x4 = PBL_FLAG_TAIL_REPEAT
        nrb = (N + 15) // 16
        max_nch = int(16 * K * sal_frac / 16) + 16
        return _lib.PblLayer(0x1000, None, N, K, (K + 511) // 512, G, nrb, flags | (1 if G > 1 else 0), max_nch, 0)

    def ws(l, M):
        return L.pbl_linear_workspace_bytes(C.byref(l), M)

    q = layer(4096, 4096, 0.1)
    assert [ws(q, M) > 0 for M in (1, 2, 4, 9, 32)] == [False, False, False, True, True]      # <= 4 tokens: one GEMV pass
    down = layer(4096, 11008, 0.1)
    assert ws(down, 2) == 0                                   # two tokens still fit a pass with two workgroups per CU (72 KB)
    assert ws(down, 9) > 0 and ws(down, 32) > 0
    ffn = layer(13824, 5120, 0.2)
    assert ws(ffn, 3) == 0 and ws(ffn, 4) > 0                 # 4 tokens would need 89 KB of LDS: matrix-core kernel
    tiny = layer(768, 2048, 0.1)
    assert ws(tiny, 8) == 0 and ws(tiny, 9) > 0               # few records: GEMV up to 8 tokens, both are latency bound
    assert ws(layer(768, 768, 0.1), 9) == 0                   # three slabs only: the matrix-core kernel runs unsplit, no scratch
    odd = layer(4096, 4100, 0.1)
    assert ws(odd, 32) == 0                                   # K % 8 != 0: the matrix-core kernel does not apply
    grp = layer(4096, 4096, 0.1, G=32)
    assert ws(grp, 2) == 0 and ws(grp, 3) > 0                 # column groups: two tokens per GEMV pass
    assert ws(layer(4096, 4096, 0.1, flags=0x4), 32) == 0     # a version-1 style blob without the slab index


def test_fp16_checkpoint_entries_must_be_recognisable_in_the_kernel_tile():
    """The matrix-core kernels keep fl16((-scale) (q - zero)) in their tile and tell entries from empty positions by "any bit
    set" (pbl_host.cpp: sal16_storable).  The packer therefore codes a zero-valued salient only with a code whose negated
    product is -0 (or a negative underflow), never +0; rows with scale 0 keep their salients as exceptions; and
    pbl_blob_describe rejects a blob in which such an entry was forged."""
    N, K = 16, 512
    W = np.full((N, K), 0.5, np.float32)
    W[:, 1::2] = -0.5
    hi, lo = np.full(N, 0.5, np.float32), np.full(N, -0.5, np.float32)
    ss, sz = np.full(N, 0.25, np.float32), np.full(N, 4.0, np.float32)
    sal = np.zeros((N, K), np.uint8)
    ss[1], sz[1] = 1e-9, 5.3            # every product underflows to +-0 in fp16; the zero point is no integer
    ss[2] = 0.0                         # degenerate quantizer: nothing can be coded
    for r in (0, 1, 2):
        W[r, 40], W[r, 41] = 0.0, 0.0
        sal[r, 40] = sal[r, 41] = 1
    p = pack_dense(W, hi, lo, ss, sz, sal, sal_f16=True)
    np.testing.assert_array_equal(F.decode(p.blob.numpy()), W)                  # exact either way
    assert p.nexc == 2 and p.nnz == 4                                           # row 2: exceptions; rows 0, 1: coded
    # the codes chosen for row 1 sit ABOVE the zero point (negated product negative -> -0), never below it
    blob = p.blob.numpy().copy()
    h = F.read_header(blob)
    rb = blob[h["rb_off_pos"]: h["rb_off_pos"] + 32].view(np.uint32).reshape(2, 4)
    off = int(rb[0, 0]) * 16
    nfull, ntail = int(rb[0, 1]), int(rb[0, 2])
    nch = nfull + ntail
    sal_off = off + 512 + h["P"] * 1024
    a128 = lambda v: (v + 127) & ~127                                           # noqa: E731  (PBL_SAL_*_OFF of include/pbl.h)
    code_off = sal_off + a128(2 * nch) + a128(16 * nch)
    codes = blob[code_off: code_off + 16 * nch].reshape(nch, 16)
    crow_off = code_off + a128(16 * nch) + ((ntail + 15) & ~15)
    crow = blob[crow_off: crow_off + nch]
    row1 = [int(c) for ch in range(nch) if crow[ch] == 1 for c in codes[ch][:2]]
    assert row1 and all(c >= 6 for c in row1), row1
    # forge: move row 1's first code below the zero point -> its negated product would be +0
    ch1 = int(np.flatnonzero(crow == 1)[0])
    blob[code_off + 16 * ch1] = 5
    layer = _lib.PblLayer()
    assert _lib.lib().pbl_blob_describe(blob.ctypes.data, blob.size, C.byref(layer)) == _lib.PBL_ERR_BAD_BLOB
