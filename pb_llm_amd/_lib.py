"""ctypes binding of libpbl.so (C ABI in include/pbl.h).

There is deliberately NO fallback: if the shared library is missing the import
of anything that computes raises, so a GPU test can never pass on a silent CPU
path.  Build it with `python -c "import __graft_entry__ as g; g.build()"`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PBL_LIB: an alternative build of the same library (A/B runs of kernel variants from tools/); the default is the in-tree one
LIB_PATH = os.environ.get("PBL_LIB") or os.path.join(_HERE, "libpbl.so")

PBL_MAX_TOKENS_PER_LAUNCH = 4
PBL_FUSED_INLINE_MAX = 4
PBL_FLAG_HAS_GROUPS = 0x1
PBL_FLAG_SAL_F16 = 0x2
PBL_FLAG_TAIL_REPEAT = 0x4
PBL_FLAG_SLABS = 0x8
PBL_PACK_COUNT_WORDS = 56
PBL_DTYPE_F32, PBL_DTYPE_F16, PBL_DTYPE_BF16 = 0, 1, 2
(PBL_OK, PBL_ERR_INVALID_ARG, PBL_ERR_BAD_BLOB, PBL_ERR_UNSUPPORTED, PBL_ERR_MISALIGNED, PBL_ERR_CAPACITY, PBL_ERR_LAUNCH,
 PBL_ERR_NOT_REPRESENTABLE) = (0, -1, -2, -3, -4, -5, -6, -7)


class PblLayer(C.Structure):
    """struct pbl_layer (include/pbl.h)."""
    _fields_ = [("blob", C.c_void_p), ("bias", C.c_void_p),
                ("N", C.c_uint32), ("K", C.c_uint32), ("P", C.c_uint32), ("G", C.c_uint32),
                ("NRB", C.c_uint32), ("flags", C.c_uint32), ("max_nch", C.c_uint32),
                ("max_nexc", C.c_uint32)]


class PblBlobHeader(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("N", C.c_uint32), ("K", C.c_uint32),
                ("P", C.c_uint32), ("G", C.c_uint32), ("NRB", C.c_uint32), ("flags", C.c_uint32),
                ("max_nch", C.c_uint32), ("max_nexc", C.c_uint32), ("nnz", C.c_uint64),
                ("nexc", C.c_uint64), ("blob_bytes", C.c_uint64), ("rb_off_pos", C.c_uint32),
                ("reserved", C.c_uint32 * 3)]


class PblError(RuntimeError):
    pass


_lib = None

EXPORTS = ["pbl_status_string", "pbl_version", "pbl_pack_dense_f32", "pbl_pack_dev_count", "pbl_pack_dev_write", "pbl_blob_describe",
           "pbl_unpack_dense_f32", "pbl_unpack_dev", "pbl_gemv_lds_bytes", "pbl_linear_f16", "pbl_linear_f16_ws", "pbl_gemm_mfma_f16",
           "pbl_gemm_mfma_f16_ws", "pbl_mfma_workspace_bytes", "pbl_linear_workspace_bytes", "pbl_gemv_f16_grouped", "pbl_gemv_f16_fused", "pbl_gemv_f16_fused_host", "pbl_gemm_f16", "pbl_gemm_f16_ex", "pbl_gemm_f16_ws", "pbl_gemm_workspace_bytes", "pbl_gemm_list_bytes", "pbl_gemm_prepare", "pbl_gemm_f16_prepared",
           "pbl_gemm_image_stats_bytes", "pbl_gemm_image_stats", "pbl_gemm_image_bytes", "pbl_gemm_image_build", "pbl_gemm_image_build_residual", "pbl_act_f32_join3", "pbl_x_fragment_bytes", "pbl_x_to_fragments", "pbl_gemm_f16_image_xf", "pbl_gemm_f16_image",
           "pbl_gemm_small_image_workspace_bytes", "pbl_gemm_small_image_ws", "pbl_gemm_small_image_act", "pbl_act_f32_split", "pbl_act_f32_join",
           "pbl_qat_workspace_bytes", "pbl_qat_scale", "pbl_qat_wsim", "pbl_qat_wgrad",
           "pbl_prep_workspace_bytes", "pbl_kth_pair", "pbl_outlier_mask", "pbl_quant8_rows", "pbl_high_calibrate", "pbl_gptq_block",
           "pbl_p2p_buffer_bytes", "pbl_comm_alloc", "pbl_comm_free", "pbl_ipc_export", "pbl_ipc_open", "pbl_ipc_close",
           "pbl_p2p_allreduce_f32", "pbl_p2p_allreduce_f32_dev", "pbl_p2p_buffer_bytes_world", "pbl_p2p_check",
           "pbl_linear_f16_push", "pbl_p2p_reduce_f32_dev", "pbl_linear_push_max_tokens",
           "pbl_gemm_f16_image_ex", "pbl_act_bf16_prepare", "pbl_act_finish",
           "pbl_linear_bf16", "pbl_gemv_bf16_fused_host", "pbl_gemm_image_workspace_bytes", "pbl_gemm_image_plan",
           "pbl_gemm_f16_image_ws"]


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so; load it first so libpbl.so binds to the SAME
    # HIP runtime instance (streams / device pointers are per-runtime).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PblError(f"{LIB_PATH} not found: the HIP extension is not built "
                       "(run __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    L.pbl_status_string.restype = C.c_char_p
    L.pbl_status_string.argtypes = [C.c_int]
    L.pbl_version.restype = C.c_int
    L.pbl_pack_dense_f32.restype = C.c_int
    L.pbl_pack_dense_f32.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp, vp, u32, vp, sz, C.POINTER(sz)]
    L.pbl_pack_dev_count.restype = C.c_int
    L.pbl_pack_dev_count.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp, vp, u32, vp, vp]
    L.pbl_pack_dev_write.restype = C.c_int
    L.pbl_pack_dev_write.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp, vp, u32, vp, vp, C.c_uint64, u32, u32, C.c_uint64, C.c_uint64, vp, vp]
    L.pbl_blob_describe.restype = C.c_int
    L.pbl_blob_describe.argtypes = [vp, sz, C.POINTER(PblLayer)]
    L.pbl_unpack_dense_f32.restype = C.c_int
    L.pbl_unpack_dense_f32.argtypes = [vp, sz, vp]
    L.pbl_unpack_dev.restype = C.c_int
    L.pbl_unpack_dev.argtypes = [C.POINTER(PblLayer), vp, C.c_int, vp]
    L.pbl_gemv_lds_bytes.restype = sz
    L.pbl_gemv_lds_bytes.argtypes = [C.POINTER(PblLayer), C.c_int]
    L.pbl_linear_f16.restype = C.c_int
    L.pbl_linear_f16.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp]
    L.pbl_gemm_mfma_f16.restype = C.c_int
    L.pbl_gemm_mfma_f16.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp]
    L.pbl_linear_f16_ws.restype = C.c_int
    L.pbl_linear_f16_ws.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, sz, vp]
    L.pbl_gemm_mfma_f16_ws.restype = C.c_int
    L.pbl_gemm_mfma_f16_ws.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, sz, vp]
    L.pbl_mfma_workspace_bytes.restype = sz
    L.pbl_mfma_workspace_bytes.argtypes = [C.POINTER(PblLayer), C.c_int]
    L.pbl_linear_workspace_bytes.restype = sz
    L.pbl_linear_workspace_bytes.argtypes = [C.POINTER(PblLayer), C.c_int]
    L.pbl_gemv_f16_grouped.restype = C.c_int
    L.pbl_gemv_f16_grouped.argtypes = [vp, vp, vp, C.c_int, C.c_int, u32, u32, u32, u32, C.c_int, C.c_int, vp]
    L.pbl_gemm_f16.restype = C.c_int
    L.pbl_gemm_f16.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, vp]
    L.pbl_gemm_f16_ex.restype = C.c_int
    L.pbl_gemm_f16_ex.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp]
    L.pbl_gemm_f16_ws.restype = C.c_int
    L.pbl_gemm_f16_ws.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, sz, vp]
    L.pbl_gemm_workspace_bytes.restype = sz
    L.pbl_gemm_workspace_bytes.argtypes = [C.POINTER(PblLayer), C.c_int]
    L.pbl_gemm_list_bytes.restype = sz
    L.pbl_gemm_list_bytes.argtypes = [C.POINTER(PblLayer)]
    L.pbl_gemm_prepare.restype = C.c_int
    L.pbl_gemm_prepare.argtypes = [C.POINTER(PblLayer), vp, sz, vp]
    L.pbl_gemm_f16_prepared.restype = C.c_int
    L.pbl_gemm_f16_prepared.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, sz, vp]
    L.pbl_gemm_image_stats_bytes.restype = sz
    L.pbl_gemm_image_stats_bytes.argtypes = [C.POINTER(PblLayer)]
    L.pbl_gemm_image_stats.restype = C.c_int
    L.pbl_gemm_image_stats.argtypes = [C.POINTER(PblLayer), vp, vp]
    L.pbl_gemm_image_bytes.restype = sz
    L.pbl_gemm_image_bytes.argtypes = [C.POINTER(PblLayer), vp]
    L.pbl_gemm_image_build.restype = C.c_int
    L.pbl_gemm_image_build.argtypes = [C.POINTER(PblLayer), vp, vp, vp, sz, vp]
    L.pbl_gemm_image_build_residual.restype = C.c_int
    L.pbl_gemm_image_build_residual.argtypes = [C.POINTER(PblLayer), vp, vp, vp, sz, vp]
    L.pbl_x_fragment_bytes.restype = sz
    L.pbl_x_fragment_bytes.argtypes = [C.c_int, u32]
    L.pbl_x_to_fragments.restype = C.c_int
    L.pbl_x_to_fragments.argtypes = [vp, C.c_int, u32, sz, vp, vp]
    L.pbl_gemm_f16_image_xf.restype = C.c_int
    L.pbl_gemm_f16_image_xf.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, vp, sz, vp, vp, sz, vp]
    L.pbl_act_f32_join3.restype = C.c_int
    L.pbl_act_f32_join3.argtypes = [vp, C.c_int, vp, C.c_float, vp, vp, C.c_int, u32, vp, C.c_int, vp]
    L.pbl_gemm_f16_image.restype = C.c_int
    L.pbl_gemm_f16_image.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, sz, vp, vp]
    L.pbl_gemm_small_image_workspace_bytes.restype = sz
    L.pbl_gemm_small_image_workspace_bytes.argtypes = [C.POINTER(PblLayer), C.c_int]
    L.pbl_gemm_small_image_ws.restype = C.c_int
    L.pbl_gemm_small_image_ws.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, sz, vp, vp, sz, vp]
    L.pbl_act_f32_split.restype = C.c_int
    L.pbl_act_f32_split.argtypes = [vp, C.c_int, u32, sz, vp, vp, vp]
    L.pbl_act_f32_join.restype = C.c_int
    L.pbl_act_f32_join.argtypes = [vp, vp, vp, C.c_int, u32, vp, C.c_int, vp]
    L.pbl_gemm_small_image_act.restype = C.c_int
    L.pbl_gemm_small_image_act.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, vp, sz, vp, vp, sz, vp]
    L.pbl_gemv_f16_fused.restype = C.c_int
    L.pbl_gemv_f16_fused.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, u32, u32, u32, u32, C.c_int, C.c_int, vp]
    L.pbl_gemv_f16_fused_host.restype = C.c_int
    L.pbl_gemv_f16_fused_host.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, u32, u32, u32, u32, C.c_int, C.c_int, vp]
    L.pbl_qat_workspace_bytes.restype = sz
    L.pbl_qat_workspace_bytes.argtypes = []
    L.pbl_qat_scale.restype = C.c_int
    L.pbl_qat_scale.argtypes = [vp, C.c_int, vp, sz, vp, vp, vp]
    L.pbl_qat_wsim.restype = C.c_int
    L.pbl_qat_wsim.argtypes = [vp, C.c_int, vp, vp, C.c_float, vp, C.c_int, sz, vp]
    L.pbl_qat_wgrad.restype = C.c_int
    L.pbl_qat_wgrad.argtypes = [vp, C.c_int, vp, vp, C.c_float, C.c_int, sz, vp]
    L.pbl_prep_workspace_bytes.restype = sz
    L.pbl_prep_workspace_bytes.argtypes = []
    L.pbl_kth_pair.restype = C.c_int
    L.pbl_kth_pair.argtypes = [vp, C.c_int, sz, C.c_uint64, C.c_uint64, vp, vp, vp]
    L.pbl_outlier_mask.restype = C.c_int
    L.pbl_outlier_mask.argtypes = [vp, C.c_int, sz, vp, vp, vp]
    L.pbl_quant8_rows.restype = C.c_int
    L.pbl_quant8_rows.argtypes = [vp, C.c_int, u32, u32, vp, vp, vp]
    L.pbl_high_calibrate.restype = C.c_int
    L.pbl_high_calibrate.argtypes = [vp, u32, u32, C.c_float, vp, vp, vp]
    L.pbl_gptq_block.restype = C.c_int
    L.pbl_gptq_block.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, vp]
    L.pbl_p2p_buffer_bytes.restype = sz
    L.pbl_p2p_buffer_bytes.argtypes = [sz]
    L.pbl_comm_alloc.restype = C.c_int
    L.pbl_comm_alloc.argtypes = [sz, C.POINTER(vp)]
    L.pbl_comm_free.restype = C.c_int
    L.pbl_comm_free.argtypes = [vp]
    L.pbl_ipc_export.restype = C.c_int
    L.pbl_ipc_export.argtypes = [vp, vp]
    L.pbl_ipc_open.restype = C.c_int
    L.pbl_ipc_open.argtypes = [vp, C.POINTER(vp)]
    L.pbl_ipc_close.restype = C.c_int
    L.pbl_ipc_close.argtypes = [vp]
    L.pbl_p2p_allreduce_f32.restype = C.c_int
    L.pbl_p2p_allreduce_f32.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, sz, sz, u32, vp]
    L.pbl_p2p_allreduce_f32_dev.restype = C.c_int
    L.pbl_p2p_allreduce_f32_dev.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, vp, sz, sz, vp]
    L.pbl_p2p_buffer_bytes_world.restype = sz
    L.pbl_p2p_buffer_bytes_world.argtypes = [sz, C.c_int]
    L.pbl_p2p_check.restype = C.c_int
    L.pbl_p2p_check.argtypes = [vp]
    L.pbl_linear_f16_push.restype = C.c_int
    L.pbl_linear_f16_push.argtypes = [C.POINTER(PblLayer), vp, C.c_int, C.POINTER(vp), C.c_int, C.c_int, sz, vp]
    L.pbl_p2p_reduce_f32_dev.restype = C.c_int
    L.pbl_p2p_reduce_f32_dev.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, vp, sz, sz, u32, vp]
    L.pbl_linear_push_max_tokens.restype = C.c_int
    L.pbl_linear_push_max_tokens.argtypes = [C.POINTER(PblLayer)]
    L.pbl_gemm_f16_image_ex.restype = C.c_int
    L.pbl_gemm_f16_image_ex.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, vp, sz, vp, vp]
    L.pbl_act_bf16_prepare.restype = C.c_int
    L.pbl_act_bf16_prepare.argtypes = [vp, C.c_int, u32, sz, vp, vp, vp]
    L.pbl_linear_bf16.restype = C.c_int
    L.pbl_linear_bf16.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp]
    L.pbl_gemv_bf16_fused_host.restype = C.c_int
    L.pbl_gemv_bf16_fused_host.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, u32, u32, u32, u32, C.c_int, C.c_int, vp]
    L.pbl_gemm_image_workspace_bytes.restype = sz
    L.pbl_gemm_image_workspace_bytes.argtypes = [C.POINTER(PblLayer), C.c_int]
    L.pbl_gemm_image_plan.restype = C.c_int
    L.pbl_gemm_image_plan.argtypes = [C.POINTER(PblLayer), C.c_int, vp]
    L.pbl_gemm_f16_image_ws.restype = C.c_int
    L.pbl_gemm_f16_image_ws.argtypes = [C.POINTER(PblLayer), vp, vp, C.c_int, C.c_int, vp, vp, sz, vp, vp, sz, vp]
    L.pbl_act_finish.restype = C.c_int
    L.pbl_act_finish.argtypes = [vp, vp, vp, C.c_int, u32, vp, C.c_int, vp]
    _lib = L
    return L


NATIVE_PATH = os.path.join(_HERE, "libpbl_torch.so")
_native = None


def native_linear():
    """`torch.ops.pbllm_native.linear` (csrc/pbl_torch.cpp): the decode-regime forward as ONE native call -- checks, output and
    workspace allocation, stream lookup and the launch in C++ instead of ~25 us of interpreter work around a ctypes call.
    None when the dispatcher is not built or disabled (PBL_NATIVE=0): the ctypes path then serves the same kernels."""
    global _native
    if _native is None:
        _native = False
        if os.environ.get("PBL_NATIVE", "1") != "0" and os.path.exists(NATIVE_PATH) and not os.environ.get("PBL_LIB"):
            import torch
            lib()                                    # libpbl.so first: the dispatcher links against it
            try:
                torch.ops.load_library(NATIVE_PATH)
                _native = torch.ops.pbllm_native.linear
            except (OSError, RuntimeError, AttributeError) as e:   # stale / ABI-incompatible build (torch upgrade, missing libtorch_hip)
                import warnings
                warnings.warn(f"pb_llm_amd: native dispatcher {NATIVE_PATH} did not load ({e}); using the ctypes path "
                              "(same kernels, ~6 us more host time per call).  Rebuild with __graft_entry__.build(force=True)")
    return _native or None


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().pbl_status_string(status).decode()
        raise PblError(f"libpbl {what}: {msg} ({status})")
