// pbl_torch.cpp -- native dispatcher of the packed forward: `torch.ops.pbllm_native.linear`.
//
// north_star asks for "a PyTorch-ROCm C++/HIP extension with the same nn.Linear-compatible signature"; rounds 1-2 reached the
// C ABI (include/pbl.h) through ctypes from Python, which costs ~25-30 us of interpreter work per eager call (descriptor
// struct, workspace query, two allocations, stream lookup, the ctypes marshalling) -- next to a 1-6 us kernel.  This file is
// the whole decode-regime forward (rows <= 32, fp16 activations) as ONE native call: checks, output and workspace
// allocation through ATen's caching allocator (stream ordered, graph safe), the current HIP stream, pbl_linear_f16_ws.
// The callers it serves: every `module(x)` of the reference's eval loops (qat/run_qat.py:45-66, utils.py:103-123,
// gptq_pb/eval_ppl_utils.py:55-64).  Other regimes (rows > 32, fp32 / bf16 activations) stay in pb_llm_amd/quant.py.
// Host code only: built with g++ against libtorch and libpbl.so (__graft_entry__.build()).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/pbl.h"

namespace {

at::Tensor pbl_native_linear(const at::Tensor& blob, const c10::optional<at::Tensor>& bias, const at::Tensor& x, int64_t N, int64_t K,
                             int64_t P, int64_t G, int64_t NRB, int64_t flags, int64_t max_nch, int64_t max_nexc, bool out_f32) {
    TORCH_CHECK(x.is_cuda() && blob.is_cuda(), "pbllm_native.linear: GPU tensors only (the HIP kernels are the only compute path)");
    TORCH_CHECK(x.scalar_type() == at::kHalf, "pbllm_native.linear: fp16 activations");
    TORCH_CHECK(x.dim() >= 1 && x.size(-1) == K, "pbllm_native.linear: in_features mismatch: x has ", x.size(-1), ", layer has ", K);
    TORCH_CHECK(blob.device() == x.device(), "pbllm_native.linear: packed weight and input are on different devices");
    const c10::DeviceGuard guard(x.device());        // a process that drives several GPUs: launch where x lives
    const at::Tensor xc = x.reshape({-1, K}).contiguous();
    const int64_t M = xc.size(0);
    TORCH_CHECK(M >= 1 && M <= 32, "pbllm_native.linear: 1..32 rows (decode / small batch); larger batches go through pb_llm_amd.quant");
    pbl_layer L;
    L.blob = blob.data_ptr();
    L.bias = nullptr;
    if (bias.has_value() && bias->defined()) {
        TORCH_CHECK(bias->scalar_type() == at::kFloat && bias->is_contiguous() && bias->numel() == N, "pbllm_native.linear: fp32 bias [N]");
        L.bias = bias->data_ptr<float>();
    }
    L.N = uint32_t(N); L.K = uint32_t(K); L.P = uint32_t(P); L.G = uint32_t(G); L.NRB = uint32_t(NRB);
    L.flags = uint32_t(flags); L.max_nch = uint32_t(max_nch); L.max_nexc = uint32_t(max_nexc);
    std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
    shape.back() = N;
    at::Tensor y = at::empty(shape, x.options().dtype(out_f32 ? at::kFloat : at::kHalf));
    const size_t nb = M > 1 ? pbl_linear_workspace_bytes(&L, int(M)) : 0;      // one token is always one GEMV pass
    at::Tensor ws;
    if (nb) ws = at::empty({int64_t(nb)}, x.options().dtype(at::kByte));
    hipStream_t st = c10::hip::getCurrentHIPStream(x.device().index()).stream();
    const int rc = pbl_linear_f16_ws(&L, xc.data_ptr(), y.data_ptr(), int(M), out_f32 ? 1 : 0, nb ? ws.data_ptr() : nullptr, nb, st);
    TORCH_CHECK(rc == PBL_OK, "libpbl linear: ", pbl_status_string(rc), " (", rc, ")");
    return y;
}

}  // namespace

TORCH_LIBRARY(pbllm_native, m) {
    m.def("linear(Tensor blob, Tensor? bias, Tensor x, int N, int K, int P, int G, int NRB, int flags, int max_nch, int max_nexc, bool out_f32) -> Tensor");
}
TORCH_LIBRARY_IMPL(pbllm_native, CUDA, m) { m.impl("linear", pbl_native_linear); }
