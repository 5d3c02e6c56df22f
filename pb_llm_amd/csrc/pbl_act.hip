// pbl_act.hip -- bf16 activations for the packed kernels, without a host round trip.
//
// The reference runs its QAT and its evaluation loops under bf16 (qat/run_qat.py:120 `bf16=True`; HF LLaMA checkpoints are
// bf16), i.e. F.linear(x_bf16, w, b) (quant/outlier_quantizer.py:101-106).  The packed kernels multiply fp16 tiles with fp32
// accumulation.  bf16 -> fp16 is exact inside fp16's range (8 significand bits into 11), and a token's whole row of x may be
// scaled by a power of two without changing a bit of the products' significands, so:
//
//   pbl_act_bf16_prepare   per token t: s_t = 2^max(0, exponent(amax_t) - 14); xh[t, :] = fp16(x[t, :] / s_t) (exact for every
//                          value above fp16's subnormal range after scaling: what is lost is > 2^-38 below the token's
//                          maximum); scale[t] = s_t.  A token that holds inf / NaN cannot be scaled: its row becomes the
//                          INDICATOR row  finite -> 0, +-inf -> +-1, NaN -> NaN  with scale[t] = +inf, so that
//                          y = (W . indicator) * inf  reproduces F.linear's non-finite pattern -- +inf / -inf by the sign of the
//                          weight an infinity meets, NaN where that weight is 0, NaN rows for NaN inputs (the one deviation:
//                          several infinities in ONE token whose products disagree in sign give +-inf by the weights' sum where
//                          the reference gives NaN).
//   pbl_act_finish         y_out[t, r] = cast(y_f32[t, r] * scale[t] + bias[r]): the small-batch kernels (<= 64 rows) write fp32,
//                          this turns it into the caller's dtype.  The GEMM-regime kernel applies scale and cast in its own
//                          epilogue (pbl_gemm_f16_image_ex).
// Both are plain streaming kernels (a few KB to a few MB): no host synchronisation, identical behaviour eagerly and under hipGraph
// capture -- round 4 checked the range on the HOST (one device -> host sync per call, impossible under capture, where it fell back
// to a NaN row for any non-finite token).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

constexpr int ACT_THREADS = 256;

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }

// fp32 -> bf16 bits, round to nearest even, NaN stays NaN (the +0x7FFF carry would turn an all-ones NaN into -0)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x0040u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// max over the workgroup of a non-negative uint32 key (fp32 |x| bit patterns order like unsigned integers; inf / NaN on top)
__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, uint32_t* s_red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = uint32_t(__shfl_xor(int(v), off, 64));
        v = o > v ? o : v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    uint32_t m = 0;
#pragma unroll
    for (int w = 0; w < ACT_THREADS / 64; ++w) m = s_red[w] > m ? s_red[w] : m;
    return m;
}

// one workgroup per token row
__global__ __launch_bounds__(ACT_THREADS) void act_bf16_prepare_kernel(const uint16_t* __restrict__ x, uint32_t K, size_t ldx,
                                                                       uint16_t* __restrict__ xh, float* __restrict__ scale) {
    __shared__ uint32_t s_red[ACT_THREADS / 64];
    const size_t t = blockIdx.x;
    const uint16_t* src = x + t * ldx;
    uint16_t* dst = xh + t * size_t(K);
    const int tid = threadIdx.x;
    const bool vec = (K & 7u) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;      // (dst rows: K % 8 == 0 and a 16-byte base)
    // pass 1: the largest |x| bit pattern of the row (bf16 << 16 = the fp32 pattern)
    uint32_t mx = 0;
    if (vec) {
        const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
        for (uint32_t i = tid; i < (K >> 3); i += ACT_THREADS) {
            const u32x4 v = s4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = (v[j] << 16) & 0x7FFFFFFFu, hi = v[j] & 0x7FFF0000u;
                mx = lo > mx ? lo : mx;
                mx = hi > mx ? hi : mx;
            }
        }
    } else {
        for (uint32_t i = tid; i < K; i += ACT_THREADS) {
            const uint32_t a = (uint32_t(src[i]) << 16) & 0x7FFFFFFFu;
            mx = a > mx ? a : mx;
        }
    }
    mx = block_max_u32(mx, s_red);
    const bool finite = mx < 0x7F800000u;
    // 2^-e, e = max(0, exponent(amax) - 14): amax / 2^e < 2^15 <= fp16's largest finite value
    const int eb = int(mx >> 23) - 127 - 14;
    const int e = eb > 0 ? eb : 0;
    const float down = __builtin_bit_cast(float, uint32_t(127 - e) << 23);
    if (tid == 0) scale[t] = finite ? __builtin_bit_cast(float, uint32_t(127 + e) << 23) : __builtin_inff();
    auto conv = [&](uint32_t b) -> uint32_t {                  // one bf16 -> one fp16, as bits
        const float f = bf16_bits_to_f32(b);
        float r;
        if (finite) r = f * down;                              // exact: a power of two (down to fp16's subnormals, where the cast rounds)
        else {
            const uint32_t a = (b << 16) & 0x7FFFFFFFu;
            r = a > 0x7F800000u ? f : (a == 0x7F800000u ? ((b & 0x8000u) ? -1.f : 1.f) : 0.f);
        }
        return uint32_t(__builtin_bit_cast(uint16_t, _Float16(r)));
    };
    if (vec && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
        u32x4* d4 = reinterpret_cast<u32x4*>(dst);
        for (uint32_t i = tid; i < (K >> 3); i += ACT_THREADS) {
            const u32x4 v = s4[i];
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = conv(v[j] & 0xFFFFu) | (conv(v[j] >> 16) << 16);
            d4[i] = o;
        }
    } else {
        for (uint32_t i = tid; i < K; i += ACT_THREADS) dst[i] = uint16_t(conv(src[i]));
    }
}

template <int OUT>       // PBL_DTYPE_F32 / _F16 / _BF16
__global__ __launch_bounds__(ACT_THREADS) void act_finish_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                                 const float* __restrict__ bias, uint32_t N, size_t MN, void* __restrict__ out,
                                                                 const float* __restrict__ y2,        // y2: a second term added to y first (pbl_act_f32_join)
                                                                 const float* __restrict__ y3, float c3) {   // y3: a third term, times c3 (pbl_act_f32_join3)
    const size_t i = (size_t(blockIdx.x) * ACT_THREADS + threadIdx.x) * 4;
    if (i >= MN) return;
    auto put = [&](size_t j, float v) {
        if (OUT == PBL_DTYPE_F32) static_cast<float*>(out)[j] = v;
        else if (OUT == PBL_DTYPE_F16) static_cast<_Float16*>(out)[j] = _Float16(v);
        else static_cast<uint16_t*>(out)[j] = uint16_t(f32_to_bf16_bits(v));
    };
    if ((N & 3u) == 0) {                                        // four elements of ONE token row, 16-byte aligned
        const size_t t = i / N;
        const uint32_t r = uint32_t(i - t * N);
        const float s = scale ? scale[t] : 1.f;
        v4f v = *reinterpret_cast<const v4f*>(y + i);
        if (y2) v += *reinterpret_cast<const v4f*>(y2 + i);
        if (y3) { const v4f w = *reinterpret_cast<const v4f*>(y3 + i); for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(w[k], c3, v[k]); }
        v4f b = {0.f, 0.f, 0.f, 0.f};
        if (bias) b = *reinterpret_cast<const v4f*>(bias + r);
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = scale ? __builtin_fmaf(v[k], s, b[k]) : v[k] + b[k];
        if (OUT == PBL_DTYPE_F32) *reinterpret_cast<v4f*>(static_cast<float*>(out) + i) = v4f{o[0], o[1], o[2], o[3]};
        else {
            uint2 pk;
            if (OUT == PBL_DTYPE_F16) {
                pk.x = uint32_t(__builtin_bit_cast(uint16_t, _Float16(o[0]))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(o[1]))) << 16);
                pk.y = uint32_t(__builtin_bit_cast(uint16_t, _Float16(o[2]))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(o[3]))) << 16);
            } else {
                pk.x = f32_to_bf16_bits(o[0]) | (f32_to_bf16_bits(o[1]) << 16);
                pk.y = f32_to_bf16_bits(o[2]) | (f32_to_bf16_bits(o[3]) << 16);
            }
            *reinterpret_cast<uint2*>(static_cast<uint16_t*>(out) + i) = pk;
        }
        return;
    }
    for (size_t j = i; j < MN && j < i + 4; ++j) {
        const size_t t = j / N;
        const float b = bias ? bias[j - t * N] : 0.f;
        float v = y2 ? y[j] + y2[j] : y[j];
        if (y3) v = __builtin_fmaf(y3[j], c3, v);
        put(j, scale ? __builtin_fmaf(v, scale[t], b) : v + b);
    }
}

// fp32 activations as two fp16 terms (the kernels are linear in x: W x = W hi + W lo, accumulated in fp32), written as rows [0, M)
// and [M, 2 M) of xh.  One workgroup per token row.  With `scale` (round 6; ADVICE r5: |x| >= 65520 gave hi = inf, lo = -inf and a
// NaN row where F.linear(x_f32, ..) is finite) the row is first scaled by the power of two pbl_act_bf16_prepare uses,
// s = 2^max(0, exponent(amax) - 14):  hi = fp16(x / s), lo = fp16(x / s - hi), scale[t] = s -- the same bits as the unscaled form
// whenever amax < 2^15; a token that holds inf / NaN becomes its indicator row in hi (finite -> 0, +-inf -> +-1, NaN -> NaN), lo = 0,
// scale[t] = +inf, so that (W . indicator) * inf has F.linear's pattern.  Without `scale`: hi = fp16(x), lo = fp16(x - hi).
__global__ __launch_bounds__(ACT_THREADS) void act_f32_split_kernel(const float* __restrict__ x, uint32_t K, size_t ldx, size_t MK,
                                                                    uint16_t* __restrict__ xh, float* __restrict__ scale) {
    __shared__ uint32_t s_red[ACT_THREADS / 64];
    const size_t t = blockIdx.x;
    const float* src = x + t * ldx;
    uint16_t* dhi = xh + t * size_t(K);
    uint16_t* dlo = dhi + MK;
    const int tid = threadIdx.x;
    const bool vec = (K & 3u) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dhi) & 7) == 0 && (MK & 3) == 0;
    bool finite = true;
    float down = 1.f;
    if (scale) {
        uint32_t mx = 0;
        if (vec) {
            for (uint32_t i = tid; i < (K >> 2); i += ACT_THREADS) {
                const u32x4 v = reinterpret_cast<const u32x4*>(src)[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t a = v[j] & 0x7FFFFFFFu; mx = a > mx ? a : mx; }
            }
        } else {
            for (uint32_t i = tid; i < K; i += ACT_THREADS) { const uint32_t a = __builtin_bit_cast(uint32_t, src[i]) & 0x7FFFFFFFu; mx = a > mx ? a : mx; }
        }
        mx = block_max_u32(mx, s_red);
        finite = mx < 0x7F800000u;
        const int eb = int(mx >> 23) - 127 - 14;
        const int e = eb > 0 ? eb : 0;
        down = __builtin_bit_cast(float, uint32_t(127 - e) << 23);
        if (tid == 0) scale[t] = finite ? __builtin_bit_cast(float, uint32_t(127 + e) << 23) : __builtin_inff();
    }
    auto two = [&](float v, uint32_t& hi, uint32_t& lo) {
        if (finite) {
            const float w = v * down;                            // exact (a power of two; fp32 subnormals of a row whose maximum is > 2^15 aside)
            const _Float16 h = _Float16(w);
            hi = uint32_t(__builtin_bit_cast(uint16_t, h));
            lo = uint32_t(__builtin_bit_cast(uint16_t, _Float16(w - float(h))));
        } else {
            const uint32_t u = __builtin_bit_cast(uint32_t, v), a = u & 0x7FFFFFFFu;
            const float r = a > 0x7F800000u ? v : (a == 0x7F800000u ? ((u >> 31) ? -1.f : 1.f) : 0.f);
            hi = uint32_t(__builtin_bit_cast(uint16_t, _Float16(r)));
            lo = 0u;
        }
    };
    if (vec) {
        for (uint32_t i = tid; i < (K >> 2); i += ACT_THREADS) {
            const v4f v = reinterpret_cast<const v4f*>(src)[i];
            uint32_t h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) two(v[k], h[k], l[k]);
            reinterpret_cast<uint2*>(dhi)[i] = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
            reinterpret_cast<uint2*>(dlo)[i] = uint2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
        }
        return;
    }
    for (uint32_t i = tid; i < K; i += ACT_THREADS) {
        uint32_t h, l;
        two(src[i], h, l);
        dhi[i] = uint16_t(h); dlo[i] = uint16_t(l);
    }
}

}  // namespace

// x [M, K] bf16 (device, rows ldx elements apart) -> x_f16 [M, K] fp16 contiguous (K % 8 == 0 and a 16-byte base make the rows
// 16-byte aligned: what the matrix-core kernels want) + tok_scale [M] fp32: see the header of this file.  One small kernel.
extern "C" int pbl_act_bf16_prepare(const void* x_bf16, int M, uint32_t K, size_t ldx, void* x_f16, float* tok_scale, void* stream) {
    if (!x_bf16 || !x_f16 || !tok_scale || M < 1 || K < 1 || ldx < K) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x_bf16) & 1) || (reinterpret_cast<uintptr_t>(x_f16) & 1) || (reinterpret_cast<uintptr_t>(tok_scale) & 3))
        return PBL_ERR_MISALIGNED;
    const uint16_t* x = static_cast<const uint16_t*>(x_bf16);
    uint16_t* xh = static_cast<uint16_t*>(x_f16);
    void* argv[] = {&x, &K, &ldx, &xh, &tok_scale};
    return hipLaunchKernel(reinterpret_cast<const void*>(act_bf16_prepare_kernel), dim3(uint32_t(M)), dim3(ACT_THREADS), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// y_out [M, N] (out_dtype: PBL_DTYPE_F32 / _F16 / _BF16) = cast(y_f32 [M, N] * tok_scale[t] + bias[r]); tok_scale and bias may be
// NULL (no scaling / no bias).  y_out may alias y_f32 only for PBL_DTYPE_F32.  One small kernel.
extern "C" int pbl_act_finish(const float* y_f32, const float* tok_scale, const float* bias, int M, uint32_t N, void* y_out, int out_dtype,
                              void* stream) {
    if (!y_f32 || !y_out || M < 1 || N < 1) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(y_f32) & 15) || (reinterpret_cast<uintptr_t>(y_out) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
        return PBL_ERR_MISALIGNED;
    const void* k = out_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_F32>)
                  : out_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_F16>)
                  : out_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_BF16>) : nullptr;
    if (!k) return PBL_ERR_INVALID_ARG;
    size_t MN = size_t(M) * N;
    const float* y2 = nullptr;
    const float* y3 = nullptr;
    float c3 = 0.f;
    void* argv[] = {&y_f32, &tok_scale, &bias, &N, &MN, &y_out, &y2, &y3, &c3};
    return hipLaunchKernel(k, dim3(uint32_t((MN + 4 * ACT_THREADS - 1) / (4 * ACT_THREADS))), dim3(ACT_THREADS), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// fp32 activations (the reference's fp32-only module classes, quant/quantizer.py:78-80,175-177, and QAT's fp32 master weights,
// utils.py:34-36): x [M, K] fp32 (rows ldx elements apart) -> x_f16 [2 M, K] fp16 contiguous, rows [0, M) = the high term, rows
// [M, 2 M) the low term (see act_f32_split_kernel).  tok_scale [M] (device fp32; NULL: the unscaled round-5 form, which overflows
// for |x| >= 65520): the per-token power of two the row was divided by (+inf: a token with inf / NaN).  The packed kernels run ONCE
// over the 2 M rows with an fp32 result; pbl_act_f32_join adds the halves and multiplies the scale back.
extern "C" int pbl_act_f32_split(const float* x, int M, uint32_t K, size_t ldx, void* x_f16, float* tok_scale, void* stream) {
    if (!x || !x_f16 || M < 1 || K < 1 || ldx < K) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(x_f16) & 1) || (reinterpret_cast<uintptr_t>(tok_scale) & 3)) return PBL_ERR_MISALIGNED;
    uint16_t* xh = static_cast<uint16_t*>(x_f16);
    size_t MK = size_t(M) * K;
    void* argv[] = {&x, &K, &ldx, &MK, &xh, &tok_scale};
    return hipLaunchKernel(reinterpret_cast<const void*>(act_f32_split_kernel), dim3(uint32_t(M)), dim3(ACT_THREADS), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// y_out [M, N] (out_dtype) = cast((y_f32[t, r] + y_f32[M + t, r]) * tok_scale[t] + bias[r]) for y_f32 [2 M, N]: the two terms of
// pbl_act_f32_split added in fp32, the token's scale (NULL: 1) multiplied back, the bias once.  One small kernel.
extern "C" int pbl_act_f32_join(const float* y_f32, const float* tok_scale, const float* bias, int M, uint32_t N, void* y_out, int out_dtype, void* stream) {
    if (!y_f32 || !y_out || M < 1 || N < 1) return PBL_ERR_INVALID_ARG;
    size_t MN = size_t(M) * N;
    if ((reinterpret_cast<uintptr_t>(y_f32) & 15) || (reinterpret_cast<uintptr_t>(y_out) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) ||
        (reinterpret_cast<uintptr_t>(tok_scale) & 3))
        return PBL_ERR_MISALIGNED;
    const void* k = out_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_F32>)
                  : out_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_F16>)
                  : out_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_BF16>) : nullptr;
    if (!k) return PBL_ERR_INVALID_ARG;
    const float* y2 = y_f32 + MN;
    const float* y3 = nullptr;
    float c3 = 0.f;
    void* argv[] = {&y_f32, &tok_scale, &bias, &N, &MN, &y_out, &y2, &y3, &c3};
    return hipLaunchKernel(k, dim3(uint32_t((MN + 4 * ACT_THREADS - 1) / (4 * ACT_THREADS))), dim3(ACT_THREADS), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// y_out [M, N] (out_dtype) = cast((y_f32[t, r] (+ y_f32[M + t, r] with two_terms) + lo_scale * y_lo[t, r]) * tok_scale[t] + bias[r]):
// the terms of an fp32-grid layer multiplied from its two images (pbl_gemm_image_build / _build_residual; lo_scale = 2^-12) -- y_f32
// [2 M, N] for fp32 activations (pbl_act_f32_split's two fp16 terms through the ordinary image) or [M, N] for fp16 / scaled bf16
// activations, y_lo [M, N] the HIGH activation term through the residual image.  tok_scale and bias may be NULL.  One small kernel.
extern "C" int pbl_act_f32_join3(const float* y_f32, int two_terms, const float* y_lo, float lo_scale, const float* tok_scale, const float* bias, int M,
                                 uint32_t N, void* y_out, int out_dtype, void* stream) {
    if (!y_f32 || !y_lo || !y_out || M < 1 || N < 1) return PBL_ERR_INVALID_ARG;
    size_t MN = size_t(M) * N;
    if ((reinterpret_cast<uintptr_t>(y_f32) & 15) || (reinterpret_cast<uintptr_t>(y_lo) & 15) || (reinterpret_cast<uintptr_t>(y_out) & 15) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) || (reinterpret_cast<uintptr_t>(tok_scale) & 3) || (two_terms && (MN & 3)))
        return PBL_ERR_MISALIGNED;
    const void* k = out_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_F32>)
                  : out_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_F16>)
                  : out_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(act_finish_kernel<PBL_DTYPE_BF16>) : nullptr;
    if (!k) return PBL_ERR_INVALID_ARG;
    const float* y2 = two_terms ? y_f32 + MN : nullptr;
    void* argv[] = {&y_f32, &tok_scale, &bias, &N, &MN, &y_out, &y2, &y_lo, &lo_scale};
    return hipLaunchKernel(k, dim3(uint32_t((MN + 4 * ACT_THREADS - 1) / (4 * ACT_THREADS))), dim3(ACT_THREADS), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}
