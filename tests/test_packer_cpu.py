"""CPU tests of the host logic: the C-ABI library loads and exports every symbol of
include/pbl.h, and the PBL1 packer round-trips -- checked against the independent
numpy decoder in oracle/pb_format_ref.py.  No compute kernels are called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import pb_format_ref as FR
from oracle import pb_oracle as O
from pb_llm_amd import _lib, synth
from pb_llm_amd.packing import PackedWeight, infer_code_grid, infer_levels, pack_dense

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "pbl.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(pbl_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in pbl.h but not exported"
    assert set(_lib.EXPORTS) == declared
    assert L.pbl_version() == 2
    assert L.pbl_status_string(-3).decode() == "unsupported shape or option"


def test_struct_sizes_match_header():
    assert C.sizeof(_lib.PblBlobHeader) == 80
    assert C.sizeof(_lib.PblLayer) == 48


def _ptq_case(N, K, low_frac=0.9, gs=-1, seed=3, metric="magnitude"):
    W = synth.llm_weight(N, K, seed=seed, heavy_tail=True)
    hd = None
    if metric == "hessian":
        hd = (1.0 + 20.0 * (synth.uniform01(K, seed, 9) < 0.02)).astype(np.float32)
    mask = O.ptq_low_mask(W, low_frac, metric, hd, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    return W, mask, r


@pytest.mark.parametrize("N,K,gs", [(64, 512, -1), (48, 768, -1), (40, 1000, -1), (33, 640, 128), (16, 512, 256)])
def test_pack_roundtrip_fp32_exact(N, K, gs):
    W, mask, r = _ptq_case(N, K, gs=gs)
    Wfq = r["W_fq"]
    G = 1 if gs == -1 else K // gs
    hi = (r["scale"] + r["mean"]).reshape(G, N).T
    lo = (-r["scale"] + r["mean"]).reshape(G, N).T
    p = pack_dense(Wfq, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8))
    assert (p.N, p.K, p.G) == (N, K, G)
    assert p.nexc == 0, "every salient value is on the HighQuantizer grid"
    assert p.nnz == int((~mask).sum())
    dec = FR.decode(p.blob.numpy())
    np.testing.assert_array_equal(dec, Wfq)             # independent decoder
    np.testing.assert_array_equal(p.unpack().numpy(), Wfq)  # library unpacker


def test_pack_any_input_is_exact_via_exceptions():
    rng = np.random.default_rng(0)
    W = rng.standard_normal((20, 700)).astype(np.float32)   # no structure at all
    hi, lo = infer_levels(W)
    ss, sz = infer_code_grid(W, hi, lo)
    p = pack_dense(W, hi, lo, ss, sz)
    np.testing.assert_array_equal(FR.decode(p.blob.numpy()), W)
    np.testing.assert_array_equal(p.unpack().numpy(), W)
    assert p.nexc > 0


def test_pack_sign_zero_and_empty_salient():
    W = synth.llm_weight(32, 512, seed=5)
    W[3, 7] = 0.0
    s = np.sign(W).astype(np.float32)
    one = np.ones((32, 1), np.float32)
    p = pack_dense(s, one, -one, np.ones(32, np.float32), np.zeros(32, np.float32))
    assert p.nnz == 1 and p.nexc == 0           # the sign(0)==0 entry is a code-0 entry
    np.testing.assert_array_equal(FR.decode(p.blob.numpy()), s)
    p2 = pack_dense(np.where(s == 0, 1, s), one, -one)   # no salient params at all
    assert p2.nnz == 0 and p2.max_nch == 0


def test_packer_sets_tail_repeat_flag():
    """PBL_FLAG_TAIL_REPEAT (include/pbl.h): the MFMA kernel writes all 16 entries of every chunk, so tail
    padding must repeat the last entry; oracle/pb_format_ref.decode asserts the rule on every tail chunk."""
    W = synth.llm_weight(40, 1100, seed=9)
    mask = O.ptq_low_mask(W, 0.83, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"], r["hzero"],
                   (~mask).astype(np.uint8))
    assert p.flags & _lib.PBL_FLAG_TAIL_REPEAT
    assert FR.stats(p.blob.numpy())["ntail"] > 0 if "ntail" in FR.stats(p.blob.numpy()) else True
    np.testing.assert_array_equal(FR.decode(p.blob.numpy()), r["W_fq"])


def test_pack_large_column_gaps_and_dense_rows():
    N, K = 16, 2048
    W = np.full((N, K), -0.5, np.float32)
    W[:, ::2] = 0.5
    hi = np.full((N, 1), 0.5, np.float32)
    lo = -hi
    ss = np.full(N, 0.01, np.float32)
    sz = np.full(N, 100.0, np.float32)
    W[0, [3, 900, 901, 2000]] = ss[0] * (np.array([7, 250, 0, 255], np.float32) - 100)   # gaps > 255
    W[1, :] = ss[1] * (np.arange(K) % 256 - 100).astype(np.float32)                      # fully salient row
    W[2, 5] = 123.456                                                                    # off-grid -> exception
    p = pack_dense(W, hi, lo, ss, sz)
    dec = FR.decode(p.blob.numpy())
    np.testing.assert_array_equal(dec, W)
    np.testing.assert_array_equal(p.unpack().numpy(), W)
    assert p.nexc == 1


def test_infer_structure_from_dense_checkpoint():
    """from a flattened dense matrix (no mask, no quantizer state) the levels and the
    code grid are re-discovered; result must still decode exactly."""
    W, mask, r = _ptq_case(64, 1024, seed=8)
    Wfq = r["W_fq"]
    hi, lo = infer_levels(Wfq)
    np.testing.assert_array_equal(hi[:, 0], (r["scale"] + r["mean"]).reshape(-1))
    np.testing.assert_array_equal(lo[:, 0], (-r["scale"] + r["mean"]).reshape(-1))
    ss, sz = infer_code_grid(Wfq, hi, lo)
    p = pack_dense(Wfq, hi, lo, ss, sz)
    np.testing.assert_array_equal(FR.decode(p.blob.numpy()), Wfq)
    assert p.nexc <= 0.02 * p.nnz


def test_fp16_checkpoint_packs_as_codes_not_exceptions():
    """gptq_pb writes W_fq back as fp16 (gptq.py:182): salient values are fl16(scale*(q-zero)).
    With PBL_FLAG_SAL_F16 they stay 1-byte codes and unpack bit-exactly."""
    W, mask, r = _ptq_case(64, 1024, seed=13)
    W16 = r["W_fq"].astype(np.float16)
    hi, lo = infer_levels(W16.astype(np.float32), -1, mask)
    p = pack_dense(W16.astype(np.float32), hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=True)
    assert p.flags & _lib.PBL_FLAG_SAL_F16
    assert p.nexc <= 2 and p.nnz >= int((~mask).sum()) - 2
    np.testing.assert_array_equal(FR.decode(p.blob.numpy()), W16.astype(np.float32))
    np.testing.assert_array_equal(p.unpack().numpy(), W16.astype(np.float32))
    # without the flag the same matrix needs an 8-byte exception per salient weight
    p0 = pack_dense(W16.astype(np.float32), hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8))
    assert p0.nexc > 0.5 * int((~mask).sum()) and p.nbytes < 0.5 * p0.nbytes
    # flattened checkpoint, nothing but the fp16 matrix: structure re-inferred through the rounding
    from pb_llm_amd.quant import PBLinear
    layer = PBLinear.from_dense(torch.from_numpy(W16))
    np.testing.assert_array_equal(layer.weight.cpu().numpy(), W16)
    assert layer.packed.nexc <= 0.05 * layer.packed.nnz


def test_bad_arguments_and_blobs():
    L = _lib.lib()
    sz = C.c_size_t(0)
    assert L.pbl_pack_dense_f32(None, 1, 1, 1, None, None, None, None, None, 0, None, 0, C.byref(sz)) == -1
    W = np.zeros((16, 512), np.float32)
    hi = np.ones((16, 3), np.float32)
    with pytest.raises(_lib.PblError, match="multiples of 128"):       # K % G != 0 / groupsize not a multiple of 128
        pack_dense(W, hi, -hi)
    with pytest.raises(_lib.PblError, match="16 bits"):   # column indices are 16 bit with pre-doubled byte steps: K <= 32767
        pack_dense(np.zeros((1, 32768), np.float32), np.ones((1, 1), np.float32), -np.ones((1, 1), np.float32))
    junk = torch.zeros(256, dtype=torch.uint8)
    with pytest.raises(_lib.PblError):
        PackedWeight.from_blob(junk)
    out = np.zeros(4, np.float32)
    assert L.pbl_unpack_dense_f32(junk.data_ptr(), junk.numel(), out.ctypes.data) == -2


def test_algorithmic_bytes_matches_survey_formula():
    W, mask, r = _ptq_case(64, 512)
    p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"], r["hzero"],
                   (~mask).astype(np.uint8))
    N, K, nnz = 64, 512, int((~mask).sum())
    want = N * K // 8 + 2 * nnz + 4 * N + 8 * N + 4 * (N + 1) + 2 * K + 2 * N
    assert p.algorithmic_bytes(1) == want


def test_forward_on_cpu_fails_loudly():
    from pb_llm_amd.quant import PBLinear
    W, mask, r = _ptq_case(32, 512)
    layer = PBLinear.from_dense(torch.from_numpy(r["W_fq"]), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])
    with pytest.raises(_lib.PblError):
        layer(torch.zeros(1, 512, dtype=torch.float16))


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/pbl.h is the drop-in boundary: it must compile as C99 (no C++-isms, no torch types) and a C program must
    link against libpbl.so and call the host-side entry points (pack -> describe -> unpack round trip)."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "rt.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "pbl.h"
int main(void) {
    enum { N = 20, K = 640 };
    static float W[N * K], hi[N], lo[N], ss[N], sz[N], back[N * K];
    static unsigned char sal[N * K];
    for (int r = 0; r < N; ++r) { hi[r] = 0.5f + r; lo[r] = -0.25f - r; ss[r] = 0.125f; sz[r] = 3.0f; }
    for (int i = 0; i < N * K; ++i) {
        int r = i / K;
        sal[i] = (i % 7) == 0;
        W[i] = sal[i] ? ss[r] * ((float)(i % 251) - sz[r]) : ((i % 3) ? hi[r] : lo[r]);
    }
    size_t need = 0;
    if (pbl_pack_dense_f32(W, N, K, 1, hi, lo, ss, sz, sal, 0, NULL, 0, &need) != PBL_OK || !need) return 1;
    void* blob = malloc(need);
    if (pbl_pack_dense_f32(W, N, K, 1, hi, lo, ss, sz, sal, 0, blob, need, &need) != PBL_OK) return 2;
    pbl_layer L;
    if (pbl_blob_describe(blob, need, &L) != PBL_OK || L.N != N || L.K != K || !(L.flags & PBL_FLAG_TAIL_REPEAT)) return 3;
    if (pbl_unpack_dense_f32(blob, need, back) != PBL_OK) return 4;
    for (int i = 0; i < N * K; ++i) if (back[i] != W[i]) return 5;
    printf("ok %s v%d %zu bytes\n", pbl_status_string(PBL_OK), pbl_version(), need);
    return 0;
}
''')
    exe = tmp_path / "rt"
    libdir = os.path.join(repo, "pb_llm_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(repo, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lpbl", "-L/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.startswith("ok ")
