// pbl_torch.cpp -- the packed forward as a native torch operator: `torch.ops.pbllm_native.linear`.
//
// north_star asks for "a PyTorch-ROCm C++/HIP extension with the same nn.Linear-compatible signature".  The boundary stays the
// C ABI (include/pbl.h); this file is the torch side of it in C++: ONE operator that does everything `module(x)` needs for any
// row count and activation dtype --
//   * decode / small batch (<= 32 rows): pbl_linear_f16_ws (GEMV passes or the matrix-core kernel, routed by the library), or --
//     from 5 rows, when the caller hands the layer's GEMM image over -- the small-batch kernel over the image
//     (pbl_gemm_small_image_ws);
//     bf16 activations as one fp16 pass over a per-token power-of-two-scaled copy made ON THE DEVICE (pbl_act_bf16_prepare /
//     pbl_act_finish, round 5: no host sync, +-inf / NaN as F.linear gives them, the same eagerly and under capture), fp32
//     activations as two fp16 terms;
//   * GEMM regime (> 32 rows: prefill, the reference's perplexity loops gptq_pb/eval_ppl_utils.py:55-64): the hand-written
//     kernel over the layer's GEMM image (pbl_gemm_f16_image_ex) when the caller hands one over, pbl_gemm_f16_ws otherwise --
//     for fp16, bf16 AND fp32 activations of every fp16-exact layer (round 5; backends "auto" / "fused"); pbl_unpack_dev + a
//     library GEMM only for backend "library", for "tuned" without an image, and for fp32-grid layers;
//   * a Meta kernel (shapes and dtypes without touching a GPU: torch.compile, fake tensors) and an Autograd kernel (the packed
//     weight is frozen, dx = dy @ W with W re-unpacked in the backward -- what the reference's fake-quant nn.Linear gives
//     prompt tuning / input-gradient analysis).
// Round 3 had the <= 32-row fp16 case only (7.7 us per eager call instead of 13.7 through ctypes); everything else went through
// Python (pb_llm_amd/quant.py), which now only keeps the ctypes route as the fallback for variant libraries (PBL_LIB) and
// PBL_NATIVE=0.  The callers it serves: every `module(x)` of the reference's loops (qat/run_qat.py:45-66, utils.py:103-123,
// gptq_pb/eval_ppl_utils.py:55-64).  Host code only: built with g++ against libtorch and libpbl.so (__graft_entry__.build()).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <c10/hip/HIPStream.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/pbl.h"

namespace {

constexpr int64_t MFMA_MAX = 32;        // rows the packed small-batch kernels take (pb_llm_amd/quant.py: MFMA_MAX)
constexpr int64_t GEMM_THRESHOLD = 12;  // ... and where layers the matrix-core kernel refuses switch to the dense path
constexpr int64_t SMALL_IMAGE_MAX = 64; // rows the small-batch kernel over the GEMM image takes (the image is read once for all of them)
constexpr int64_t SMALL_IMAGE_MIN = 5;  // rows from which the small-batch kernel over the GEMM image beats the one over the records (pb_llm_amd/quant.py)

pbl_layer make_layer(const at::Tensor& blob, const c10::optional<at::Tensor>& bias, int64_t N, int64_t K, int64_t P, int64_t G,
                     int64_t NRB, int64_t flags, int64_t max_nch, int64_t max_nexc, bool with_bias) {
    pbl_layer L;
    L.blob = blob.data_ptr();
    L.bias = nullptr;
    if (with_bias && bias.has_value() && bias->defined()) {
        TORCH_CHECK(bias->scalar_type() == at::kFloat && bias->is_contiguous() && bias->numel() == N, "pbllm_native.linear: fp32 bias [N]");
        L.bias = bias->data_ptr<float>();
    }
    L.N = uint32_t(N); L.K = uint32_t(K); L.P = uint32_t(P); L.G = uint32_t(G); L.NRB = uint32_t(NRB);
    L.flags = uint32_t(flags); L.max_nch = uint32_t(max_nch); L.max_nexc = uint32_t(max_nexc);
    return L;
}

// layers the matrix-core kernel (<= 32 rows) takes: K % 8 == 0, slab index, column groups of a power of two >= 128
bool mfma_ok(int64_t K, int64_t G, int64_t flags) {
    const int64_t gs = K / G;
    return (G == 1 || (gs >= 128 && (gs & (gs - 1)) == 0 && gs * G == K)) && K % 8 == 0 && (flags & PBL_FLAG_SLABS);
}
// layers pbl_gemm_f16_* take: K % 8 == 0, slab index + repeat-padded tails, groups of k * 128 columns
bool fused_ok(int64_t K, int64_t G, int64_t flags) {
    const int64_t need = PBL_FLAG_SLABS | PBL_FLAG_TAIL_REPEAT;
    if (K % 8 || (flags & need) != need) return false;
    return G == 1 || (K % G == 0 && (K / G) % 128 == 0);
}

void check(int rc, const char* what) { TORCH_CHECK(rc == PBL_OK, "libpbl ", what, ": ", pbl_status_string(rc), " (", rc, ")"); }

hipStream_t stream_of(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// the layer's GEMM image as the caller handed it over (pb_llm_amd/quant.py keeps it with the packed weight)
struct ImageRef {
    const void* data = nullptr;
    size_t bytes = 0;
    std::vector<uint32_t> geom;
    explicit operator bool() const { return data != nullptr; }
};

// <= 32 rows of fp16 x [M, K] -> y [M, N] (fp16 / fp32); scratch from the caching allocator (stream ordered, graph safe).  With
// the layer's GEMM image and SMALL_IMAGE_MIN rows or more: the small-batch kernel over the image (pbl_gemm_small_image_ws).
// pbl_gemm_small_image_ws into y (<= 64 rows); false: a layer the kernel does not take
bool run_small_image(const pbl_layer& L, const at::Tensor& xc, at::Tensor& y, int64_t M, bool f32, const ImageRef& img) {
    const size_t nbi = pbl_gemm_small_image_workspace_bytes(&L, int(M));
    at::Tensor wsi;
    if (nbi) wsi = at::empty({int64_t(nbi)}, xc.options().dtype(at::kByte));
    const int rc = pbl_gemm_small_image_ws(&L, xc.data_ptr(), y.data_ptr(), int(M), f32 ? 1 : 0, img.data, img.bytes, img.geom.data(),
                                           nbi ? wsi.data_ptr() : nullptr, nbi, stream_of(xc));
    TORCH_CHECK(rc == PBL_OK || rc == PBL_ERR_UNSUPPORTED, "libpbl gemm_small_image: ", pbl_status_string(rc), " (", rc, ")");
    return rc == PBL_OK;
}

// bf16 activations, 5 - 64 rows over the image: pbl_gemm_small_image_act (scale, bias and cast inside the K splits' reduce: one
// launch fewer than kernel + pbl_act_finish).  xh / tsc: prepare_bf16's; an undefined tensor: the layer runs as one split or is
// not taken -- the caller runs the fp32 kernel + finish.
at::Tensor run_small_image_act(const pbl_layer& L, const at::Tensor& xh, const at::Tensor& tsc, int64_t M, at::ScalarType out_dt, const ImageRef& img) {
    if (!img || M < SMALL_IMAGE_MIN || M > SMALL_IMAGE_MAX || (reinterpret_cast<uintptr_t>(xh.data_ptr()) & 15)) return at::Tensor();
    const size_t nbi = pbl_gemm_small_image_workspace_bytes(&L, int(M));
    if (!nbi) return at::Tensor();
    at::Tensor wsi = at::empty({int64_t(nbi)}, xh.options().dtype(at::kByte));
    at::Tensor y = at::empty({M, int64_t(L.N)}, xh.options().dtype(out_dt));
    const int rc = pbl_gemm_small_image_act(&L, xh.data_ptr(), y.data_ptr(), int(M), out_dt == at::kFloat ? PBL_DTYPE_F32 : PBL_DTYPE_BF16, tsc.data_ptr<float>(),
                                            img.data, img.bytes, img.geom.data(), wsi.data_ptr(), nbi, stream_of(xh));
    TORCH_CHECK(rc == PBL_OK || rc == PBL_ERR_UNSUPPORTED, "libpbl gemm_small_image_act: ", pbl_status_string(rc), " (", rc, ")");
    return rc == PBL_OK ? y : at::Tensor();
}

at::Tensor run_small(const pbl_layer& L, const at::Tensor& xc, int64_t M, bool f32, const ImageRef& img) {
    at::Tensor y = at::empty({M, int64_t(L.N)}, xc.options().dtype(f32 ? at::kFloat : at::kHalf));
    if (img && M >= SMALL_IMAGE_MIN && (reinterpret_cast<uintptr_t>(xc.data_ptr()) & 15) == 0 && run_small_image(L, xc, y, M, f32, img)) return y;
    const size_t nb = M > 1 ? pbl_linear_workspace_bytes(&L, int(M)) : 0;      // one token is always one GEMV pass
    at::Tensor ws;
    if (nb) ws = at::empty({int64_t(nb)}, xc.options().dtype(at::kByte));
    check(pbl_linear_f16_ws(&L, xc.data_ptr(), y.data_ptr(), int(M), f32 ? 1 : 0, nb ? ws.data_ptr() : nullptr, nb, stream_of(xc)), "linear");
    return y;
}

at::Tensor unpack(const pbl_layer& L, const at::Tensor& like, at::ScalarType dt) {
    at::Tensor W = at::empty({int64_t(L.N), int64_t(L.K)}, like.options().dtype(dt));
    pbl_layer nb = L;
    nb.bias = nullptr;
    check(pbl_unpack_dev(&nb, W.data_ptr(), dt == at::kFloat ? 1 : 0, stream_of(like)), "unpack_dev");
    return W;
}

// bf16 activations [M, K] -> (fp16 copy scaled per token by a power of two, tok_scale [M]): pbl_act_bf16_prepare, one small kernel,
// no host synchronisation (csrc/pbl_act.hip)
std::pair<at::Tensor, at::Tensor> prepare_bf16(const at::Tensor& x2, int64_t M, int64_t K) {
    const at::Tensor xs = x2.stride(-1) == 1 && (M == 1 || x2.stride(0) >= K) ? x2 : x2.contiguous();
    at::Tensor xh = at::empty({M, K}, x2.options().dtype(at::kHalf));
    at::Tensor sc = at::empty({M}, x2.options().dtype(at::kFloat));
    const size_t ldx = M == 1 ? size_t(K) : size_t(xs.stride(0));       // (the stride of a size-1 dimension is arbitrary)
    check(pbl_act_bf16_prepare(xs.data_ptr(), int(M), uint32_t(K), ldx, xh.data_ptr(), sc.data_ptr<float>(), stream_of(x2)), "act_bf16_prepare");
    return {xh, sc};
}

// y_out = cast(y32 * tok_scale[t] + bias[r]) (pbl_act_finish); tok_scale / bias may be undefined
at::Tensor finish(const at::Tensor& y32, const at::Tensor& tok_scale, const float* bias, int64_t M, int64_t N, at::ScalarType out_dt) {
    at::Tensor y = at::empty({M, N}, y32.options().dtype(out_dt));
    const int dt = out_dt == at::kFloat ? PBL_DTYPE_F32 : (out_dt == at::kHalf ? PBL_DTYPE_F16 : PBL_DTYPE_BF16);
    check(pbl_act_finish(y32.data_ptr<float>(), tok_scale.defined() ? tok_scale.data_ptr<float>() : nullptr, bias, int(M), uint32_t(N), y.data_ptr(), dt,
                         stream_of(y32)), "act_finish");
    return y;
}

at::Tensor linear_cuda(const at::Tensor& blob, const c10::optional<at::Tensor>& bias, const at::Tensor& x, int64_t N, int64_t K, int64_t P,
                       int64_t G, int64_t NRB, int64_t flags, int64_t max_nch, int64_t max_nexc, bool out_f32, bool dense_f16,
                       const c10::optional<at::Tensor>& image, c10::OptionalArrayRef<int64_t> geom, c10::string_view backend,
                       bool small_image, bool split_k, bool x_fragments, const c10::optional<at::Tensor>& xfrag) {
    TORCH_CHECK(x.is_cuda() && blob.is_cuda(), "pbllm_native.linear: GPU tensors only (the HIP kernels are the only compute path)");
    const auto xt = x.scalar_type();
    TORCH_CHECK(xt == at::kHalf || xt == at::kBFloat16 || xt == at::kFloat, "pbllm_native.linear: fp16, bf16 or fp32 activations");
    TORCH_CHECK(x.dim() >= 1 && x.size(-1) == K, "pbllm_native.linear: in_features mismatch: x has ", x.size(-1), ", layer has ", K);
    TORCH_CHECK(blob.device() == x.device(), "pbllm_native.linear: packed weight and input are on different devices");
    const c10::DeviceGuard guard(x.device());        // a process that drives several GPUs: launch where x lives
    std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
    shape.back() = N;
    const at::Tensor x2 = x.reshape({-1, K});
    const int64_t M = x2.size(0);
    const auto out_dt = out_f32 ? at::kFloat : xt;
    if (M == 0) return at::zeros(shape, x.options().dtype(out_dt));
    const pbl_layer L = make_layer(blob, bias, N, K, P, G, NRB, flags, max_nch, max_nexc, true);
    pbl_layer Lnb = L;
    Lnb.bias = nullptr;
    const bool has_bias = L.bias != nullptr;
    const int64_t rows = xt == at::kFloat ? 2 * M : M;     // fp32 x runs as two fp16 terms (bf16 converts exactly)
    const bool mok = mfma_ok(K, G, flags);
    ImageRef iref;
    if (image.has_value() && image->defined() && geom.has_value()) {
        iref.data = image->data_ptr(); iref.bytes = size_t(image->numel());
        iref.geom.assign(geom->begin(), geom->end());
    }
    ImageRef isml;                                         // the image as the small-batch kernel (<= 64 rows) may use it
    if (small_image) isml = iref;
    auto dense_path = [&](at::ScalarType wdt) {
        const at::Tensor W = unpack(L, x, wdt);
        at::Tensor y = at::linear(x2.to(wdt), W, has_bias ? c10::optional<at::Tensor>(bias->to(wdt)) : c10::nullopt);
        return y.to(out_dt).reshape(shape);
    };
    // fp32 activations: x = x_hi + x_lo with both terms fp16; the kernels are linear in x, so y = W x_hi + W x_lo accumulated in
    // fp32 (bias added once), each token scaled by a power of two so that no term leaves fp16's range.  One launch each way
    // (pbl_act_f32_split / _join, csrc/pbl_act.hip) around the packed kernel.
    at::Tensor f32_scale;                                  // tok_scale of split_f32(): a power of two per token (+inf: inf / NaN inside)
    auto split_f32 = [&]() {
        const at::Tensor xs = x2.stride(-1) == 1 && (M == 1 || x2.stride(0) >= K) ? x2 : x2.contiguous();
        at::Tensor xh = at::empty({2 * M, K}, x2.options().dtype(at::kHalf));
        f32_scale = at::empty({M}, x2.options().dtype(at::kFloat));
        check(pbl_act_f32_split(xs.data_ptr<float>(), int(M), uint32_t(K), M == 1 ? size_t(K) : size_t(xs.stride(0)), xh.data_ptr(),
                                f32_scale.data_ptr<float>(), stream_of(x2)), "act_f32_split");
        return xh;
    };
    auto join_f32 = [&](const at::Tensor& yy) {
        at::Tensor y = at::empty({M, N}, yy.options().dtype(out_dt));
        const int dt = out_dt == at::kFloat ? PBL_DTYPE_F32 : (out_dt == at::kHalf ? PBL_DTYPE_F16 : PBL_DTYPE_BF16);
        check(pbl_act_f32_join(yy.data_ptr<float>(), f32_scale.data_ptr<float>(), L.bias, int(M), uint32_t(N), y.data_ptr(), dt, stream_of(yy)), "act_f32_join");
        return y.reshape(shape);
    };
    if (mok ? rows > MFMA_MAX : M >= GEMM_THRESHOLD) {
        // GEMM regime.  Round 5: every layer an fp16 checkpoint is exact for runs on the hand-written kernels whatever the
        // activation dtype ("auto" / "fused"): fp16 directly, bf16 as its per-token-scaled fp16 copy with the scale and the bf16
        // cast in the GEMM's epilogue, fp32 as two fp16 terms.  Only fp32-grid layers (the reference's fp32-only module classes,
        // quant/quantizer.py:78,175) keep the fp32 library GEMM on the unpacked weight: fp16 tiles cannot meet their 2e-5 bar.
        if (backend != "library" && dense_f16 && fused_ok(K, G, flags)) {
            at::Tensor xin, tsc;
            if (xt == at::kHalf) xin = x2.contiguous();
            else if (xt == at::kBFloat16) std::tie(xin, tsc) = prepare_bf16(x2, M, K);
            else xin = split_f32();
            const int64_t R = xin.size(0);
            const bool img = bool(iref);
            if ((reinterpret_cast<uintptr_t>(xin.data_ptr()) & 15) == 0 && (img || backend != "tuned")) {
                // the kernel's own output type: what the caller wants when it can produce it, else fp32 + pbl_act_finish / the join
                const bool direct = xt == at::kHalf || (xt == at::kBFloat16 && !out_f32 && img && R > SMALL_IMAGE_MAX);
                const bool k32 = xt == at::kHalf ? out_f32 : !direct;
                const pbl_layer& Lk = (xt == at::kHalf || direct) ? L : Lnb;     // (bias: in the kernel, or once behind it)
                at::Tensor y = at::empty({R, N}, x.options().dtype(k32 ? at::kFloat : (xt == at::kBFloat16 ? at::kBFloat16 : at::kHalf)));
                bool done = false;
                if (xt == at::kBFloat16 && img && R <= SMALL_IMAGE_MAX && small_image) {
                    const at::Tensor ya = run_small_image_act(L, xin, tsc, R, out_dt, iref);
                    if (ya.defined()) return ya.reshape(shape);
                }
                if (img && R <= SMALL_IMAGE_MAX && small_image) done = run_small_image(Lk, xin, y, R, k32, iref);   // 33 - 64 rows: one pass over the image
                if (!done && img) {
                    // a thin last round (5120-row layers at 2048 rows, short prompts) is cut off and split along K through a transient
                    // workspace (pbl_gemm_f16_image_ws); split_k == false: one launch, bit-identical to the round-3 kernel
                    const size_t wb = split_k ? pbl_gemm_image_workspace_bytes(&Lk, int(R)) : 0;
                    at::Tensor wsk;
                    if (wb) wsk = at::empty({int64_t(wb)}, x.options().dtype(at::kByte));
                    const int odt = k32 ? PBL_DTYPE_F32 : (direct && xt == at::kBFloat16 ? PBL_DTYPE_BF16 : PBL_DTYPE_F16);
                    const float* tsp = direct && xt == at::kBFloat16 ? tsc.data_ptr<float>() : nullptr;
                    if (x_fragments) {
                        // round 6: x as a fragment-major copy -- the caller's (fp16 x: kept per activation tensor, shared by q / k / v and
                        // gate / up) or made here (the scaled / split fp16 copies of bf16 / fp32 activations); no x tile through LDS
                        at::Tensor xfr;
                        if (xt == at::kHalf && xfrag.has_value() && xfrag->defined()) xfr = *xfrag;
                        else {
                            xfr = at::empty({int64_t(pbl_x_fragment_bytes(int(R), uint32_t(K)))}, x.options().dtype(at::kByte));
                            check(pbl_x_to_fragments(xin.data_ptr(), int(R), uint32_t(K), size_t(K), xfr.data_ptr(), stream_of(x)), "x_to_fragments");
                        }
                        check(pbl_gemm_f16_image_xf(&Lk, xfr.data_ptr(), y.data_ptr(), int(R), odt, tsp, iref.data, iref.bytes, iref.geom.data(),
                                                    wb ? wsk.data_ptr() : nullptr, wb, stream_of(x)), "gemm_f16_image_xf");
                    } else
                    check(pbl_gemm_f16_image_ws(&Lk, xin.data_ptr(), y.data_ptr(), int(R), odt, tsp, iref.data, iref.bytes, iref.geom.data(),
                                                wb ? wsk.data_ptr() : nullptr, wb, stream_of(x)),
                          "gemm_f16_image");
                } else if (!done) {
                    const size_t nb = pbl_gemm_workspace_bytes(&Lk, int(R));
                    at::Tensor ws;
                    if (nb) ws = at::empty({int64_t(nb)}, x.options().dtype(at::kByte));
                    check(pbl_gemm_f16_ws(&Lk, xin.data_ptr(), y.data_ptr(), int(R), k32 ? 1 : 0, nb ? ws.data_ptr() : nullptr, nb, stream_of(x)), "gemm_f16");
                }
                if (xt == at::kHalf || direct) return y.reshape(shape);
                if (xt == at::kBFloat16) return finish(y, tsc, L.bias, M, N, out_dt).reshape(shape);
                return join_f32(y);
            }
        }
        return dense_path((xt == at::kHalf && dense_f16) ? at::kHalf : at::kFloat);
    }
    if (xt == at::kHalf) return run_small(L, x2.contiguous(), M, out_f32, isml).reshape(shape);
    if (xt == at::kBFloat16) {
        // <= 32 rows: prepare (scaled fp16 copy + per-token scale, on the device) -> the packed kernels with an fp32 result ->
        // scale, bias and cast in pbl_act_finish.  The same three launches eagerly and under stream capture; +-inf / NaN inputs come
        // out as F.linear's do (csrc/pbl_act.hip).
        if (M <= PBL_MAX_TOKENS_PER_LAUNCH && G == 1) {
            // decode: ONE launch -- the GEMV's staging phase converts, its epilogue scales back and rounds to bf16 (pbl_linear_bf16:
            // the same bits as the three launches below)
            const at::Tensor xs = x2.contiguous();
            at::Tensor y = at::empty({M, N}, x.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
            const int rc = pbl_linear_bf16(&L, xs.data_ptr(), y.data_ptr(), int(M), out_f32 ? 1 : 0, stream_of(x));
            if (rc == PBL_OK) return y.reshape(shape);
            TORCH_CHECK(rc == PBL_ERR_UNSUPPORTED, "libpbl linear_bf16: ", pbl_status_string(rc), " (", rc, ")");
        }
        at::Tensor xh, tsc;
        std::tie(xh, tsc) = prepare_bf16(x2, M, K);
        const at::Tensor ya = run_small_image_act(L, xh, tsc, M, out_dt, isml);
        if (ya.defined()) return ya.reshape(shape);
        const at::Tensor y32 = run_small(Lnb, xh, M, true, isml);
        return finish(y32, tsc, L.bias, M, N, out_dt).reshape(shape);
    }
    return join_f32(run_small(Lnb, split_f32(), 2 * M, true, isml));
}

at::Tensor linear_meta(const at::Tensor& blob, const c10::optional<at::Tensor>& bias, const at::Tensor& x, int64_t N, int64_t K, int64_t P,
                       int64_t G, int64_t NRB, int64_t flags, int64_t max_nch, int64_t max_nexc, bool out_f32, bool dense_f16,
                       const c10::optional<at::Tensor>& image, c10::OptionalArrayRef<int64_t> geom, c10::string_view backend,
                       bool small_image, bool split_k, bool x_fragments, const c10::optional<at::Tensor>& xfrag) {
    TORCH_CHECK(x.dim() >= 1 && x.size(-1) == K, "pbllm_native.linear: in_features mismatch: x has ", x.size(-1), ", layer has ", K);
    std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
    shape.back() = N;
    return at::empty(shape, x.options().dtype(out_f32 ? at::kFloat : x.scalar_type()));
}

// the frozen packed weight is differentiable in x: dx = dy @ W, W re-unpacked in the backward (nothing dense is saved)
class PBLinearFn : public torch::autograd::Function<PBLinearFn> {
 public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& blob, const c10::optional<at::Tensor>& bias,
                              int64_t N, int64_t K, int64_t P, int64_t G, int64_t NRB, int64_t flags, int64_t max_nch, int64_t max_nexc, bool out_f32,
                              bool dense_f16, const c10::optional<at::Tensor>& image, c10::OptionalArrayRef<int64_t> geom, std::string backend,
                              bool small_image, bool split_k, bool x_fragments, const c10::optional<at::Tensor>& xfrag) {
        ctx->saved_data["blob"] = blob;
        ctx->saved_data["meta"] = std::vector<int64_t>{N, K, P, G, NRB, flags, max_nch, max_nexc};
        ctx->saved_data["xdt"] = int64_t(x.scalar_type());
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("pbllm_native::linear", "")
                             .typed<at::Tensor(const at::Tensor&, const c10::optional<at::Tensor>&, const at::Tensor&, int64_t, int64_t, int64_t, int64_t,
                                               int64_t, int64_t, int64_t, int64_t, bool, bool, const c10::optional<at::Tensor>&,
                                               c10::OptionalArrayRef<int64_t>, c10::string_view, bool, bool, bool, const c10::optional<at::Tensor>&)>();
        return op.call(blob, bias, x, N, K, P, G, NRB, flags, max_nch, max_nexc, out_f32, dense_f16, image, geom, backend, small_image, split_k,
                       x_fragments, xfrag);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const at::Tensor dy = grads[0];
        const at::Tensor blob = ctx->saved_data["blob"].toTensor();
        const auto m = ctx->saved_data["meta"].toIntVector();
        const auto xdt = at::ScalarType(ctx->saved_data["xdt"].toInt());
        const c10::DeviceGuard guard(dy.device());
        const pbl_layer L = make_layer(blob, c10::nullopt, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], false);
        const auto wdt = (dy.scalar_type() == at::kHalf && (m[5] & PBL_FLAG_SAL_F16)) ? at::kHalf : at::kFloat;
        const at::Tensor W = unpack(L, dy, wdt);
        std::vector<int64_t> shape(dy.sizes().begin(), dy.sizes().end());
        shape.back() = m[1];
        const at::Tensor dx = dy.reshape({-1, m[0]}).to(wdt).matmul(W).reshape(shape).to(xdt);
        torch::autograd::variable_list out(20);
        out[0] = dx;
        return out;
    }
};

at::Tensor linear_autograd(const at::Tensor& blob, const c10::optional<at::Tensor>& bias, const at::Tensor& x, int64_t N, int64_t K, int64_t P,
                           int64_t G, int64_t NRB, int64_t flags, int64_t max_nch, int64_t max_nexc, bool out_f32, bool dense_f16,
                           const c10::optional<at::Tensor>& image, c10::OptionalArrayRef<int64_t> geom, c10::string_view backend,
                           bool small_image, bool split_k, bool x_fragments, const c10::optional<at::Tensor>& xfrag) {
    return PBLinearFn::apply(x, blob, bias, N, K, P, G, NRB, flags, max_nch, max_nexc, out_f32, dense_f16, image, geom, std::string(backend),
                             small_image, split_k, x_fragments, xfrag);
}

}  // namespace

TORCH_LIBRARY(pbllm_native, m) {
    m.def("linear(Tensor blob, Tensor? bias, Tensor x, int N, int K, int P, int G, int NRB, int flags, int max_nch, int max_nexc, bool out_f32, "
          "bool dense_f16=True, Tensor? image=None, int[]? geom=None, str backend=\"auto\", bool small_image=True, bool split_k=True, "
          "bool x_fragments=False, Tensor? xfrag=None) -> Tensor");
}
TORCH_LIBRARY_IMPL(pbllm_native, CUDA, m) { m.impl("linear", linear_cuda); }
TORCH_LIBRARY_IMPL(pbllm_native, Meta, m) { m.impl("linear", linear_meta); }
TORCH_LIBRARY_IMPL(pbllm_native, Autograd, m) { m.impl("linear", linear_autograd); }
