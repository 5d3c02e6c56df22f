"""CPU tests of round 5: the ISA audits are part of the suite and are proven non-vacuous, build hygiene, the
activation-dtype entry points' argument checks."""
import ctypes as C
import importlib.util
import os
import re
import subprocess

import pytest
import torch

from pb_llm_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def audit():
    spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(REPO, "tools", "audit_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_isa_audit_passes_on_the_shipped_kernels(audit):
    """tools/audit_asm_loads.py over the generated gfx950 ISA: the round-3 GEMM kernel's asm loads (main), the image kernel's slot
    requests and its x hand-over rule (main_img: round 4 ran it from the command line only), and round 5's generic wait audit --
    vector-memory destinations, LDS-DMA pieces and LDS operations against the counted waits -- on every instantiation of the
    small-batch kernel (30 kernels: ring of slot sets, staging wave) and of the image kernel."""
    assert audit.main_img() == 0
    assert audit.main_waits(pattern="pbl_sb_img_kernel") == 0
    assert audit.main_waits(pattern="pbl_gemm_img_kernel", dma_rule=False) == 0


def test_the_audit_finds_the_vmcnt13_build_of_the_image_kernel(audit):
    """round 4 shipped `s_waitcnt vmcnt(13)` for a day: up to three x pieces of the previous step stayed in flight at the barrier
    that publishes them, 2 of 256 workgroups computed wrong rows and every parity case passed.  The same listing with the step's
    counted wait rewritten to 13 must be reported -- and the shipped wait must not."""
    asm = audit.compile_asm(audit.IMG_SRC)
    bodies = audit.img_kernel_bodies(asm)
    assert len(bodies) == 6
    for name, lines in bodies.items():
        assert audit.audit_img(lines)[0] == [], name
        lax, n = [], 0
        for i, l in enumerate(lines):
            if "s_waitcnt vmcnt(10)" in l and i and "#ASMSTART" in lines[i - 1]:
                l = l.replace("vmcnt(10)", "vmcnt(13)")
                n += 1
            lax.append(l)
        assert n >= 4, (name, n)
        problems, _ = audit.audit_img(lax)
        assert any(isinstance(p[2], list) and "in flight at the barrier" in str(p[2][0]) for p in problems), name


def test_the_audit_finds_a_one_too_lax_wait_in_the_small_batch_kernel(audit):
    """every counted wait of pbl_sb_img_kernel -- the compiler's vmcnt over the ring of slot sets, its lgkmcnt over the fragment
    reads, the staging wave's vmcnt(0) in front of the barrier that publishes the x tile, the working waves' lgkmcnt(0) in front of
    the barrier that frees it -- made ONE laxer is reported; likewise the image kernel's counted lgkmcnt(6) over its fragment reads"""
    asm = audit.compile_asm(audit.IMG_SRC)
    # (round 6: the workgroup geometry RP x KQ is a template parameter pair -- the round-4 geometry 4 x 1 at both token-block counts,
    # and the K-phase geometry 2 x 4, whose reduction adds two barriers and LDS traffic of its own)
    for pat in ("pbl_sb_img_kernelILi2ELb0ELi1ELi4ELi1E", "pbl_sb_img_kernelILi3ELb1ELi2ELi4ELi1E", "pbl_sb_img_kernelILi2ELb0ELi1ELi2ELi4E"):
        (name, lines), = audit.kernel_bodies_named(asm, pat).items()
        assert audit.audit_waits(lines)[0] == []
        for counter in ("vmcnt", "lgkmcnt"):
            k, missed = 0, []
            while True:
                lax = audit.mutate_wait(lines, k, 1, counter)
                if lax is None:
                    break
                if not audit.audit_waits(lax)[0]:
                    missed.append(k)
                k += 1
            # (the only waits that may be loosened unnoticed are redundant ones: the compiler does not see the `s_waitcnt lgkmcnt(0)`
            # the source issues from inline asm in front of a barrier and waits again behind it)
            assert k >= 8 and len(missed) <= 1, (name, counter, k, missed)
        # the staging wave: vmcnt(0) -> vmcnt(1) in front of its barrier leaves an LDS-DMA piece in flight
        hits = 0
        for i, l in enumerate(lines):
            if re.search(r"s_waitcnt vmcnt\(0\)\s*$", l.split(";")[0]):
                lax = list(lines)
                lax[i] = l.replace("vmcnt(0)", "vmcnt(1)")
                hits += any("LDS-DMA" in f[2] for f in audit.audit_waits(lax)[0])
        assert hits >= 1, name
        # a working wave that enters the barrier with a fragment read outstanding
        hits = 0
        for i, l in enumerate(lines):
            if "lgkmcnt(0)" in l and any("s_barrier" in x for x in lines[i + 1:i + 3]):
                lax = list(lines)
                lax[i] = l.replace("lgkmcnt(0)", "lgkmcnt(1)")
                hits += any("LDS operation in flight at the barrier" in f[2] for f in audit.audit_waits(lax)[0])
        assert hits >= 1, name
    (name, lines), = audit.kernel_bodies_named(asm, "pbl_gemm_img_kernelILi0ELb0ELb0E").items()
    k = n6 = 0
    while True:
        lax = audit.mutate_wait(lines, k, 1, "lgkmcnt")
        if lax is None:
            break
        assert audit.audit_waits(lax, dma_rule=False)[0], (name, k)
        k += 1
    assert k >= 12
    # round 6, the XF instantiations (x as a fragment-major copy): the MFMA waves' B fragments come by plain loads under ONE counted
    # `vmcnt(6)`, the expanding waves' fixed five-load requests under `vmcnt(15)`, the A fragments under `lgkmcnt(4)`: every one of them
    # made one laxer is reported by the generic audit
    for pat in ("pbl_gemm_img_kernelILi0ELb0ELb1E", "pbl_gemm_img_kernelILi2ELb0ELb1E"):
        (name, lines), = audit.kernel_bodies_named(asm, pat).items()
        assert audit.audit_waits(lines, dma_rule=False)[0] == []
        seen, missed = {6: 0, 15: 0}, []
        for i, l in enumerate(lines):
            m = re.search(r"s_waitcnt vmcnt\((6|15)\)\s*$", l.split(";")[0])
            if m:
                lax = list(lines)
                lax[i] = l.replace(f"vmcnt({m.group(1)})", f"vmcnt({int(m.group(1)) + 1})")
                if audit.audit_waits(lax, dma_rule=False)[0]:
                    seen[int(m.group(1))] += 1
                else:
                    missed.append((i, l.strip()))
        # (unnoticed only where the wait is redundant: a compiler-placed vmcnt(6) right in front of the prologue's vmcnt(0), and the
        # LAST step's vmcnt(15), behind which no slot set is expanded any more)
        assert seen[6] >= 16 and seen[15] >= 4 and len(missed) <= 2, (name, seen, missed)
        k = 0
        while True:
            lax = audit.mutate_wait(lines, k, 1, "lgkmcnt")
            if lax is None:
                break
            assert audit.audit_waits(lax, dma_rule=False)[0], (name, k)
            k += 1
        assert k >= 12, (name, k)


def test_the_library_builds_without_warnings():
    """hipcc -Wall over every source of libpbl.so: no diagnostics at all -- in particular no -Wpass-failed (round 4's
    amdgpu-waves-per-eu attribute promised four waves per SIMD to small-batch variants that get three)"""
    import __graft_entry__ as g
    src = os.path.join(REPO, "pb_llm_amd", "csrc", "pbl_gemm_img.hip")
    r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-Wall",
                        "-Wno-unused-function", "--cuda-device-only", "-c", src, "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr[-2000:]
    assert callable(g.build)


def test_activation_entry_points_check_their_arguments():
    L = _lib.lib()
    lay = _lib.PblLayer(None, None, 64, 1024, 2, 1, 4, 0xE, 8, 0)
    assert L.pbl_act_bf16_prepare(None, 1, 8, 8, None, None, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_act_bf16_prepare(16, 1, 8, 4, 16, 16, None) == _lib.PBL_ERR_INVALID_ARG          # ldx < K
    assert L.pbl_act_bf16_prepare(17, 1, 8, 8, 16, 16, None) == _lib.PBL_ERR_MISALIGNED
    assert L.pbl_act_finish(None, None, None, 1, 8, None, 0, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_act_finish(16, None, None, 1, 8, 16, 7, None) == _lib.PBL_ERR_INVALID_ARG        # unknown dtype
    assert L.pbl_act_finish(16, None, None, 1, 8, 20, 0, None) == _lib.PBL_ERR_MISALIGNED
    # pbl_gemm_f16_image_ex: bf16 output needs the per-token scales, the other types must not get them
    assert L.pbl_gemm_f16_image_ex(C.byref(lay), 16, 16, 4, _lib.PBL_DTYPE_BF16, None, 16, 64, 16, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_gemm_f16_image_ex(C.byref(lay), 16, 16, 4, 9, None, 16, 64, 16, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_act_f32_split(None, 1, 8, 8, 16, 16, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_act_f32_split(16, 2, 8, 4, 16, 16, None) == _lib.PBL_ERR_INVALID_ARG             # ldx < K
    assert L.pbl_act_f32_split(18, 1, 8, 8, 16, 16, None) == _lib.PBL_ERR_MISALIGNED
    assert L.pbl_act_f32_split(16, 1, 8, 8, 16, 18, None) == _lib.PBL_ERR_MISALIGNED              # tok_scale (round 6)
    assert L.pbl_act_f32_join(None, None, None, 1, 8, 16, 0, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_act_f32_join(16, None, None, 1, 8, 16, 7, None) == _lib.PBL_ERR_INVALID_ARG      # unknown dtype
    assert L.pbl_act_f32_join(16, None, 20, 1, 8, 16, 0, None) == _lib.PBL_ERR_MISALIGNED         # bias
    assert L.pbl_act_f32_join(16, 18, None, 1, 8, 16, 0, None) == _lib.PBL_ERR_MISALIGNED         # tok_scale
    # the small-batch kernel for scaled activations: a token scale and a bf16 / fp32 result, 1 - 64 rows
    assert L.pbl_gemm_small_image_act(C.byref(lay), 16, 16, 8, _lib.PBL_DTYPE_BF16, None, 16, 64, 16, None, 0, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_gemm_small_image_act(C.byref(lay), 16, 16, 8, _lib.PBL_DTYPE_F16, 16, 16, 64, 16, None, 0, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_gemm_small_image_act(C.byref(lay), 16, 16, 65, _lib.PBL_DTYPE_BF16, 16, 16, 64, 16, None, 0, None) == _lib.PBL_ERR_INVALID_ARG
    assert L.pbl_linear_push_max_tokens(None) == 0


def test_the_gemm_launch_plan_without_a_gpu():
    """pbl_gemm_image_plan (host arithmetic only; 256 CUs assumed without a device): shapes that are a whole number of rounds stay one
    launch; 5120-row layers at 2048 rows get a row tail of 8 row tiles split 4 ways behind one full round; a short prompt is split
    entirely; the workspace is KS x region x 4 bytes"""
    L = _lib.lib()

    def plan(N, K, M):
        lay = _lib.PblLayer(None, None, N, K, (K + 511) // 512, 1, (N + 15) // 16, 0xE, 8, 0)
        out = (C.c_uint64 * 6)()
        assert L.pbl_gemm_image_plan(C.byref(lay), M, out) == 0
        ws = L.pbl_gemm_image_workspace_bytes(C.byref(lay), M)
        assert ws == (out[2] * out[4] * out[5] * 4 if out[0] else 0)
        return list(out)

    for N, K in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192)):
        assert plan(N, K, 2048)[0] == 0, (N, K)
    assert plan(5120, 5120, 2048) == [2, 32, 4, 10, 2048, 1024]
    assert plan(5120, 13824, 2048) == [2, 32, 4, 27, 2048, 1024]
    assert plan(13824, 5120, 2048) == [2, 96, 2, 20, 2048, 1536]          # 864 tiles: three full rounds (96 row tiles) + 12 row tiles split in two
    assert plan(11008, 4096, 1536) == [2, 85, 8, 4, 1536, 128]            # 516 tiles: the 6 tiles beyond two rounds are cut off and split 8 ways
    p = plan(4096, 4096, 300)
    assert p[0] == 1 and p[1] == 0 and p[2] * p[3] >= 32 and p[4:] == [300, 4096]
    assert plan(512, 1024, 300)[0] == 0 and plan(4096, 4096, 4096)[0] == 0
    lay = _lib.PblLayer(None, None, 4096, 4100, 9, 1, 256, 0xE, 8, 0)                 # K % 8: no image, no plan
    assert L.pbl_gemm_image_plan(C.byref(lay), 300, (C.c_uint64 * 6)()) == _lib.PBL_ERR_UNSUPPORTED


def test_concat_rows_is_byte_surgery_that_validates():
    """packing.concat_rows: q | k | v as ONE packed layer without re-packing -- records copied behind one another, record table
    rebuilt; the result passes the structural validation (pbl_blob_describe), and both the host unpacker and the independent
    decoder read the row-wise concatenation of the parts, bit for bit; mismatched parts are refused"""
    import numpy as np
    from oracle import pb_oracle as O
    from oracle import pb_format_ref as FR
    from pb_llm_amd import synth
    from pb_llm_amd.packing import PackedWeight, concat_rows, infer_levels, pack_dense
    parts, Ws = [], []
    for i, (N, K) in enumerate([(64, 1024), (32, 1024), (40, 1024)]):       # the last part may end in a partial record
        W = synth.llm_weight(N, K, seed=i + 1, heavy_tail=True)
        mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        Wd = r["W_fq"].astype(np.float16).astype(np.float32)
        if i == 1:
            Wd[3, 77] = 0.4321                                              # an exception
        hi, lo = infer_levels(Wd, -1, mask)
        parts.append(pack_dense(Wd, hi, lo, r["hscale"], r["hzero"], (~mask).astype(np.uint8), sal_f16=True))
        Ws.append(Wd)
    m = concat_rows(parts)
    v = PackedWeight.from_blob(m.blob)
    assert (v.N, v.K, v.NRB, v.max_nch, v.max_nexc, v.nnz, v.nexc) == (136, 1024, 9, m.max_nch, 1, sum(p.nnz for p in parts), 1)
    want = np.concatenate(Ws, 0)
    assert np.array_equal(m.unpack().numpy(), want) and np.array_equal(FR.decode(m.blob.numpy()), want)
    assert concat_rows(parts[:1]).blob.equal(parts[0].blob)                 # one part: the same bytes
    with pytest.raises(_lib.PblError):
        concat_rows([parts[2], parts[0]])                                   # 40 rows in front: not a whole number of records
    other = pack_dense(Ws[0][:, :512], *infer_levels(Ws[0][:, :512], -1, None))
    with pytest.raises(_lib.PblError):
        concat_rows([parts[0], other])                                      # another in_features


def test_a_layer_that_has_run_can_be_deep_copied_and_pickled():
    """a forward caches a ctypes descriptor (and, on the GPU, the GEMM image with its stream event) on the PackedWeight: derived data,
    not state -- copy.deepcopy(model) after a forward used to fail with "ctypes objects containing pointers cannot be pickled" """
    import copy
    import pickle
    from pb_llm_amd.packing import PackedWeight
    from pb_llm_amd.quant import PBLinear
    p = PackedWeight(torch.zeros(256, dtype=torch.uint8), 16, 512, 1, 1, 1, 0xE, 8, 0, 0, 0)
    p.layer_struct(None)
    p._gemm_image = (("key",), object())
    q = copy.deepcopy(p)
    assert set(q.__dict__) == set(PackedWeight._FIELDS) and torch.equal(q.blob, p.blob) and q.blob is not p.blob
    assert pickle.loads(pickle.dumps(p)).K == 512
    import numpy as np
    from pb_llm_amd.packing import pack_dense
    W = np.where(np.arange(32 * 512).reshape(32, 512) % 3 == 0, 0.5, -0.5).astype(np.float32)
    real = pack_dense(W, np.full((32, 1), 0.5, np.float32), np.full((32, 1), -0.5, np.float32))
    lin = PBLinear(real, torch.zeros(32))
    lin.packed.layer_struct(lin.pbl_bias)
    twin = copy.deepcopy(lin)
    assert twin.packed.blob is twin.pbl_blob and "_struct" not in twin.packed.__dict__
    assert torch.equal(twin.packed.unpack(), lin.packed.unpack())


def test_every_tool_script_compiles_and_names_only_exported_entry_points():
    """tools/ holds the benches, probes and profiling jobs the numbers in profiles/ come from; most need a GPU.  What can be checked
    here: every script byte-compiles, every `pbl_*` entry point a script calls through the ctypes handle is one the library
    exports (include/pbl.h = _lib.EXPORTS, checked elsewhere), and every shell job passes `bash -n`."""
    import glob, py_compile, subprocess, tempfile
    from pb_llm_amd import _lib
    tools = os.path.join(REPO, "tools")
    debug_hooks = {"pbl_debug_force_gemm_plan", "pbl_debug_set_small_image_waves", "pbl_debug_set_small_image_plan", "pbl_debug_trace_gemm_img",
                   "pbl_debug_trace_gemm", "pbl_debug_set_gemm_split"}
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(glob.glob(os.path.join(tools, "*.py"))):
            py_compile.compile(f, cfile=os.path.join(tmp, "x.pyc"), doraise=True)
            src = open(f).read()
            for name in set(re.findall(r"\b(?:L|lib\(\)|_lib\.lib\(\))\.(pbl_[a-z0-9_]+)", src)):
                assert name in _lib.EXPORTS or name in debug_hooks, (os.path.basename(f), name)
    for f in sorted(glob.glob(os.path.join(tools, "*.sh"))):
        assert subprocess.run(["bash", "-n", f], capture_output=True).returncode == 0, f
