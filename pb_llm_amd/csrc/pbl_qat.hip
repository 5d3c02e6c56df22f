// pbl_qat.hip -- the weight-side elementwise work of a QAT step of the partially-binarized layer
// (quant/outlier_quantizer.py:83-99 forward, quant/quantizer.py:18-25 straight-through backward),
// fused into three HBM-bound streaming kernels.  The reference spends ~8 elementwise / indexing passes
// per forward on it (boolean-index gather for the scale, abs, mean, mul, sign, mul, where); here it is
//   1. pbl_qat_scale : s = mean |W| over the NON-salient entries           (read W + mask)
//   2. pbl_qat_wsim  : w_sim = mask ? W * outlier_scale : sign(W) * s      (read W + mask, write w_sim,
//                      directly in the GEMM dtype: fp32 master weights -> bf16 under autocast)
//   3. pbl_qat_wgrad : dL/dW = dL/dw_sim * (mask ? (train_outlier ? outlier_scale : 0) : s), in place
// The two GEMMs of the step (y = x w_sim^T, dX = dY w_sim, dW_sim = dY^T X) are plain library GEMMs.
// All three are deterministic: fixed grid, fixed reduction tree.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

namespace {

constexpr int QT = 256;                 // threads per workgroup
constexpr int QG = PBL_QAT_PARTIALS;    // workgroups of the reduction = partials in the workspace

template <typename T> struct Vec;       // 16-byte vector of T and its mask bytes
template <> struct Vec<float> { static constexpr int W = 4; using M = uint32_t; };
template <> struct Vec<_Float16> { static constexpr int W = 8; using M = uint64_t; };
template <> struct Vec<__hip_bfloat16> { static constexpr int W = 8; using M = uint64_t; };

template <typename T> __device__ __forceinline__ float to_f(T v) { return float(v); }
template <> __device__ __forceinline__ float to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v) { return T(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename T, int W>
struct alignas(sizeof(T) * W) Pack { T v[W]; };

__device__ __forceinline__ float sgn(float w) { return w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f); }   // torch.sign: sign(0) = 0

// ---- 1. sum |W| and count over mask == 0 ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(QT) void qat_scale_stage1(const T* __restrict__ Wp, const uint8_t* __restrict__ mask, size_t n,
                                                        double* __restrict__ part_sum, uint32_t* __restrict__ part_cnt) {
    constexpr int VW = Vec<T>::W;
    using MV = typename Vec<T>::M;
    const size_t nv = n / VW;
    double acc = 0.0;                              // fp64 throughout: the result is the correctly rounded mean
    uint32_t cnt = 0;
    for (size_t i = size_t(blockIdx.x) * QT + threadIdx.x; i < nv; i += size_t(QG) * QT) {
        const Pack<T, VW> w = reinterpret_cast<const Pack<T, VW>*>(Wp)[i];
        const MV m = reinterpret_cast<const MV*>(mask)[i];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const bool sal = (m >> (8 * e)) & 0xFF;
            acc += sal ? 0.0 : double(fabsf(to_f(w.v[e])));
            cnt += sal ? 0u : 1u;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < int(n - nv * VW)) {          // scalar tail
        const size_t i = nv * VW + threadIdx.x;
        if (!mask[i]) { acc += double(fabsf(to_f(Wp[i]))); ++cnt; }
    }
    __shared__ double ss[QT];
    __shared__ uint32_t sc[QT];
    ss[threadIdx.x] = acc; sc[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = QT / 2; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) { ss[threadIdx.x] += ss[threadIdx.x + s]; sc[threadIdx.x] += sc[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_sum[blockIdx.x] = ss[0]; part_cnt[blockIdx.x] = sc[0]; }
}

__global__ __launch_bounds__(QT) void qat_scale_stage2(const double* __restrict__ part_sum, const uint32_t* __restrict__ part_cnt,
                                                        float* __restrict__ scale_out) {
    __shared__ double ss[QT];
    __shared__ unsigned long long sc[QT];
    double a = 0.0;
    unsigned long long c = 0;
    for (int i = threadIdx.x; i < QG; i += QT) { a += part_sum[i]; c += part_cnt[i]; }
    ss[threadIdx.x] = a; sc[threadIdx.x] = c;
    __syncthreads();
    for (int s = QT / 2; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) { ss[threadIdx.x] += ss[threadIdx.x + s]; sc[threadIdx.x] += sc[threadIdx.x + s]; }
        __syncthreads();
    }
    // torch: mean over an empty selection is nan
    if (threadIdx.x == 0) scale_out[0] = sc[0] ? float(ss[0] / double(sc[0])) : __builtin_nanf("");
}

// ---- 2. w_sim ---------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(QT) void qat_wsim_kernel(const TI* __restrict__ Wp, const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ scale, float outlier_scale,
                                                       TO* __restrict__ out, size_t n) {
    constexpr int VW = Vec<TI>::W;
    using MV = typename Vec<TI>::M;
    const float s = to_f(from_f<TI>(scale[0]));    // binary_scale lives in the weight's dtype
    const size_t nv = n / VW;
    for (size_t i = size_t(blockIdx.x) * QT + threadIdx.x; i < nv; i += size_t(gridDim.x) * QT) {
        const Pack<TI, VW> w = reinterpret_cast<const Pack<TI, VW>*>(Wp)[i];
        const MV m = reinterpret_cast<const MV*>(mask)[i];
        Pack<TO, VW> o;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const float wf = to_f(w.v[e]);
            // the reference multiplies in the weight's dtype: round the product to TI first, then to TO
            const float v = ((m >> (8 * e)) & 0xFF) ? to_f(from_f<TI>(wf * outlier_scale)) : to_f(from_f<TI>(sgn(wf) * s));
            o.v[e] = from_f<TO>(v);
        }
        reinterpret_cast<Pack<TO, VW>*>(out)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < int(n - nv * VW)) {
        const size_t i = nv * VW + threadIdx.x;
        const float wf = to_f(Wp[i]);
        out[i] = from_f<TO>(mask[i] ? to_f(from_f<TI>(wf * outlier_scale)) : to_f(from_f<TI>(sgn(wf) * s)));
    }
}

// ---- 3. straight-through weight gradient, in place ---------------------------------------
template <typename T>
__global__ __launch_bounds__(QT) void qat_wgrad_kernel(T* __restrict__ g, const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ scale, float sal_coef, size_t n) {
    constexpr int VW = Vec<T>::W;
    using MV = typename Vec<T>::M;
    const float s = to_f(from_f<T>(scale[0]));
    const size_t nv = n / VW;
    for (size_t i = size_t(blockIdx.x) * QT + threadIdx.x; i < nv; i += size_t(gridDim.x) * QT) {
        Pack<T, VW> v = reinterpret_cast<Pack<T, VW>*>(g)[i];
        const MV m = reinterpret_cast<const MV*>(mask)[i];
#pragma unroll
        for (int e = 0; e < VW; ++e) v.v[e] = from_f<T>(to_f(v.v[e]) * (((m >> (8 * e)) & 0xFF) ? sal_coef : s));
        reinterpret_cast<Pack<T, VW>*>(g)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < int(n - nv * VW)) {
        const size_t i = nv * VW + threadIdx.x;
        g[i] = from_f<T>(to_f(g[i]) * (mask[i] ? sal_coef : s));
    }
}

inline int grid_for(size_t n, int vw) {
    const size_t nv = (n + vw - 1) / vw, blocks = (nv + QT - 1) / QT;
    return int(blocks < 8192 ? (blocks ? blocks : 1) : 8192);      // >> 256 CUs, grid-stride beyond
}

inline bool aligned16(const void* p) { return !(reinterpret_cast<uintptr_t>(p) & 15); }

inline int launch(const void* k, int grid, void** argv, void* stream) {
    return hipLaunchKernel(k, dim3(grid), dim3(QT), argv, 0, static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

}  // namespace

extern "C" {

size_t pbl_qat_workspace_bytes(void) { return size_t(QG) * 12; }

int pbl_qat_scale(const void* W, int w_dtype, const uint8_t* mask, size_t n, void* workspace, float* scale_out, void* stream) {
    if (!W || !mask || !workspace || !scale_out || !n) return PBL_ERR_INVALID_ARG;
    if (!aligned16(W) || !aligned16(mask) || !aligned16(workspace)) return PBL_ERR_MISALIGNED;
    double* ps = static_cast<double*>(workspace);
    uint32_t* pc = reinterpret_cast<uint32_t*>(ps + QG);
    const void* k1 = w_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(qat_scale_stage1<float>)
                   : w_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(qat_scale_stage1<_Float16>)
                   : w_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(qat_scale_stage1<__hip_bfloat16>) : nullptr;
    if (!k1) return PBL_ERR_UNSUPPORTED;
    void* a1[] = {&W, &mask, &n, &ps, &pc};
    const int rc = launch(k1, QG, a1, stream);
    if (rc != PBL_OK) return rc;
    void* a2[] = {&ps, &pc, &scale_out};
    return launch(reinterpret_cast<const void*>(qat_scale_stage2), 1, a2, stream);
}

int pbl_qat_wsim(const void* W, int w_dtype, const uint8_t* mask, const float* scale, float outlier_scale,
                 void* out, int out_dtype, size_t n, void* stream) {
    if (!W || !mask || !scale || !out || !n) return PBL_ERR_INVALID_ARG;
    if (!aligned16(W) || !aligned16(mask) || !aligned16(out)) return PBL_ERR_MISALIGNED;
    const void* k = nullptr;
    int vw = 8;
    if (w_dtype == PBL_DTYPE_F32 && out_dtype == PBL_DTYPE_F32) { k = reinterpret_cast<const void*>(qat_wsim_kernel<float, float>); vw = 4; }
    else if (w_dtype == PBL_DTYPE_F32 && out_dtype == PBL_DTYPE_F16) { k = reinterpret_cast<const void*>(qat_wsim_kernel<float, _Float16>); vw = 4; }
    else if (w_dtype == PBL_DTYPE_F32 && out_dtype == PBL_DTYPE_BF16) { k = reinterpret_cast<const void*>(qat_wsim_kernel<float, __hip_bfloat16>); vw = 4; }
    else if (w_dtype == PBL_DTYPE_F16 && out_dtype == PBL_DTYPE_F16) k = reinterpret_cast<const void*>(qat_wsim_kernel<_Float16, _Float16>);
    else if (w_dtype == PBL_DTYPE_BF16 && out_dtype == PBL_DTYPE_BF16) k = reinterpret_cast<const void*>(qat_wsim_kernel<__hip_bfloat16, __hip_bfloat16>);
    else return PBL_ERR_UNSUPPORTED;
    void* argv[] = {&W, &mask, &scale, &outlier_scale, &out, &n};
    return launch(k, grid_for(n, vw), argv, stream);
}

int pbl_qat_wgrad(void* g, int g_dtype, const uint8_t* mask, const float* scale, float outlier_scale, int train_outlier,
                  size_t n, void* stream) {
    if (!g || !mask || !scale || !n) return PBL_ERR_INVALID_ARG;
    if (!aligned16(g) || !aligned16(mask)) return PBL_ERR_MISALIGNED;
    float sal_coef = train_outlier ? outlier_scale : 0.f;
    const void* k = g_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(qat_wgrad_kernel<float>)
                  : g_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(qat_wgrad_kernel<_Float16>)
                  : g_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(qat_wgrad_kernel<__hip_bfloat16>) : nullptr;
    if (!k) return PBL_ERR_UNSUPPORTED;
    void* argv[] = {&g, &mask, &scale, &sal_coef, &n};
    return launch(k, grid_for(n, g_dtype == PBL_DTYPE_F32 ? 4 : 8), argv, stream);
}

}  // extern "C"
