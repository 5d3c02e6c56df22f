// pbl_prep.hip -- the producer immediately in front of the partially-binarized layer, on the GPU:
// BinaryXnorExceptOutliersLinear.gen_outlier_mask (quant/outlier_quantizer.py:54-81) and
// weight_quant_8bit (quant/outlier_quantizer.py:10-29).  The reference runs two torch.kthvalue over the
// whole tensor (seconds per layer on the host) plus ~10 elementwise passes; here it is
//   1. pbl_kth_pair      exact k-th smallest for two ranks at once: 3-pass MSB radix select (11+11+10 bits) over
//                        order-preserving uint32 keys, LDS histograms, no host round trip        3 reads of W
//   2. pbl_outlier_mask  mask = (w < lo) | (w > hi), strict like the reference (:69)                1 read, 1 B/elt out
//   3. pbl_quant8_rows   per-row asymmetric 8-bit fake quantisation IN PLACE with the reference's quirks
//                        (integer-rounded zero point, wrapping uint8 cast), row staged in LDS      1 read, 1 write
// Everything is integer-exact or uses unfused IEEE operations in the reference's order (__f*_rn), so the
// results are bit-identical to the reference's CPU arithmetic.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

// bit-identical to the reference's unfused torch arithmetic: HIP's default -ffp-contract=fast would turn
// q*step + zp into one v_fma_f32 with a single rounding (and does so even through the __f*_rn wrappers and a
// contract(off) pragma once they are inlined), so products that feed an add go through an opaque register move
#pragma clang fp contract(off)

namespace {

constexpr int PT = 256;
constexpr int NBINS = 2048;

template <typename T> __device__ __forceinline__ float ld_f(const T* p, size_t i) { return float(p[i]); }
template <> __device__ __forceinline__ float ld_f<__hip_bfloat16>(const __hip_bfloat16* p, size_t i) { return __bfloat162float(p[i]); }
template <typename T> __device__ __forceinline__ T rnd(float v) { return T(v); }
template <> __device__ __forceinline__ __hip_bfloat16 rnd<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <typename T> __device__ __forceinline__ float rt(float v) { return float(rnd<T>(v)); }          // round through T
template <> __device__ __forceinline__ float rt<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }
template <> __device__ __forceinline__ float rt<float>(float v) { return v; }

__device__ __forceinline__ float unfused(float prod) {
    asm volatile("" : "+v"(prod));
    return prod;
}

// order-preserving map float -> uint32 (ascending), and back
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// select state in the workspace: prefix[2], remaining rank[2], then hist[2][NBINS]
struct SelState { uint32_t prefix[2]; uint32_t krem[2]; };

__global__ void sel_init(SelState* st, uint32_t* hist, uint32_t k_lo, uint32_t k_hi) {
    for (int i = threadIdx.x; i < 2 * NBINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) { st->prefix[0] = st->prefix[1] = 0; st->krem[0] = k_lo; st->krem[1] = k_hi; }
}

// histogram of the digit (key >> shift) & (nb-1) over the elements whose higher bits equal the target's prefix
template <typename T>
__global__ __launch_bounds__(PT) void sel_hist(const T* __restrict__ W, size_t n, const SelState* __restrict__ st,
                                                uint32_t* __restrict__ hist, int shift, int bits, int first) {
    __shared__ uint32_t lh[2 * NBINS];
    for (int i = threadIdx.x; i < 2 * NBINS; i += PT) lh[i] = 0;
    __syncthreads();
    const uint32_t p0 = st->prefix[0], p1 = st->prefix[1], dm = (1u << bits) - 1u;
    const int hs = shift + bits;                                   // bits above the digit
    for (size_t i = size_t(blockIdx.x) * PT + threadIdx.x; i < n; i += size_t(gridDim.x) * PT) {
        const uint32_t key = f2key(ld_f(W, i));
        const uint32_t hi = first ? 0u : (key >> hs), d = (key >> shift) & dm;
        if (first) atomicAdd(&lh[d], 1u);                          // both ranks share the first histogram
        else {
            if (hi == p0) atomicAdd(&lh[d], 1u);
            if (hi == p1) atomicAdd(&lh[NBINS + d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NBINS; i += PT)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// one workgroup: locate each rank's bin, extend its prefix, reduce its remaining rank, clear the histograms
__global__ __launch_bounds__(PT) void sel_scan(SelState* st, uint32_t* hist, int bits, int first, int last, float* out2) {
    __shared__ uint32_t part[PT];
    __shared__ uint32_t found_bin[2], found_below[2];
    const int nb = 1 << bits, per = NBINS / PT;                    // 8 bins per thread
    for (int t = 0; t < 2; ++t) {
        const uint32_t* h = hist + (first ? 0 : t * NBINS);
        uint32_t loc = 0;
        for (int q = 0; q < per; ++q) { const int b = threadIdx.x * per + q; if (b < nb) loc += h[b]; }
        part[threadIdx.x] = loc;
        __syncthreads();
        if (threadIdx.x == 0) {                                    // 256 partials: serial is fine
            const uint32_t k = st->krem[t];
            uint32_t cum = 0;
            int owner = PT - 1;
            for (int i = 0; i < PT; ++i) { if (cum + part[i] >= k) { owner = i; break; } cum += part[i]; }
            uint32_t below = cum;
            int bin = owner * per;
            for (int q = 0; q < per; ++q) {
                const int b = owner * per + q;
                if (b >= nb) break;
                bin = b;
                if (below + h[b] >= k) break;
                below += h[b];
            }
            found_bin[t] = uint32_t(bin); found_below[t] = below;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int t = 0; t < 2; ++t) {
            st->prefix[t] = (st->prefix[t] << bits) | found_bin[t];
            st->krem[t] -= found_below[t];
            if (last) out2[t] = key2f(st->prefix[t]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NBINS; i += PT) hist[i] = 0;
}

template <typename T>
__global__ __launch_bounds__(PT) void mask_kernel(const T* __restrict__ W, size_t n, const float* __restrict__ thr,
                                                   uint8_t* __restrict__ mask) {
    const float lo = thr[0], hi = thr[1];
    for (size_t i = size_t(blockIdx.x) * PT + threadIdx.x; i < n; i += size_t(gridDim.x) * PT) {
        const float w = ld_f(W, i);
        mask[i] = (w < lo) | (w > hi);
    }
}

// weight_quant_8bit, one workgroup per row, row staged in LDS (K * 4 B <= 64 KiB)
template <typename T>
__global__ __launch_bounds__(PT) void quant8_rows_kernel(T* __restrict__ W, uint32_t K, float* __restrict__ code_scale,
                                                          float* __restrict__ code_zp) {
    extern __shared__ float row[];
    __shared__ float smin[PT], smax[PT];
    T* wr = W + size_t(blockIdx.x) * K;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (uint32_t j = threadIdx.x; j < K; j += PT) {
        const float v = ld_f(wr, j);
        row[j] = v;
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = PT / 2; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    // quant/outlier_quantizer.py:13-17: range in the weight dtype then fp32; zero point = round(min) in the weight dtype
    const float range = rt<T>(__fsub_rn(smax[0], smin[0]));
    const float zp = rt<T>(rintf(smin[0]));
    const float step = __fdiv_rn(range, 255.0f);                   // :24  w_range / 255
    if (threadIdx.x == 0) { code_scale[blockIdx.x] = step; code_zp[blockIdx.x] = zp; }
    for (uint32_t j = threadIdx.x; j < K; j += PT) {
        const float t = __fmul_rn(__fdiv_rn(rt<T>(__fsub_rn(row[j], zp)), range), 255.0f);   // :18-20
        const float r = rintf(t);
        // .type(torch.uint8) on the reference's CPU: through int64, low 8 bits; nan/inf/overflow -> 0
        const uint32_t q = (r == r && fabsf(r) < 9.0e18f) ? uint32_t(uint64_t(int64_t(r)) & 255u) : 0u;
        wr[j] = rnd<T>(unfused(float(q) * step) + zp);                                        // :24-25
    }
}

// ---- GPTQ-PB column block (gptq_pb/gptq.py:129-168) -------------------------------------------------
// The reference walks the K columns in a Python loop, ~15 small torch launches per column.  Rows are independent
// inside a 128-column block, so here ONE launch processes a whole block: a wave owns a row, lane l holds columns
// l and l + 64 of the row's block in registers, the 128 x 128 block of the upper Cholesky factor of H^-1 sits in
// LDS, and the column recurrence runs 128 wave-uniform steps (v_readlane for the pivot column):
//   q    = mask ? sign(w - mean) * scale + mean : hscale * (clamp(rint(w / hscale) + hzero, 0, maxq) - hzero)
//   err  = (w - q) / d;   loss += (w - q)^2 / d^2;   W1[:, j >= i] -= err * U1[i, j]
// in fp32 with the reference's unfused roundings.  Outputs: the block's quantised columns (into W), Err1 (for the
// trailing library GEMM  W[:, c1:] -= Err1 @ U[c0:c1, c1:]), per-row loss.  feedback == 0 is the RTN branch (:119-127).
constexpr int GB = 128;     // block size (gptq.py blocksize default)

__global__ __launch_bounds__(PT) void gptq_block_kernel(float* __restrict__ W, uint32_t N, uint32_t K, uint32_t c0, uint32_t nb,
                                                         const float* __restrict__ U, const uint8_t* __restrict__ low_mask,
                                                         const float* __restrict__ hscale, const float* __restrict__ hzero, float maxq,
                                                         const float* __restrict__ mean, const float* __restrict__ scale,
                                                         float* __restrict__ Err, float* __restrict__ losses, int feedback) {
    extern __shared__ float U1[];                   // [GB][GB], rows/cols beyond nb are zero
    for (uint32_t t = threadIdx.x; t < GB * GB; t += PT) {
        const uint32_t i = t / GB, j = t % GB;
        U1[t] = (i < nb && j < nb) ? U[size_t(c0 + i) * K + c0 + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * (PT / 64) + (threadIdx.x >> 6);
    if (r >= N) return;
    float* wrow = W + size_t(r) * K + c0;
    const uint8_t* mrow = low_mask + size_t(r) * K + c0;
    const bool in0 = uint32_t(lane) < nb, in1 = uint32_t(lane) + 64 < nb;
    float w0 = in0 ? wrow[lane] : 0.f, w1 = in1 ? wrow[lane + 64] : 0.f;
    const int m0 = in0 ? mrow[lane] : 0, m1 = in1 ? mrow[lane + 64] : 0;
    const float hs = hscale[r], hz = hzero[r], mu = mean[r], sc = scale[r];
    float q0 = 0.f, q1 = 0.f, e0 = 0.f, e1 = 0.f, loss = 0.f;
    for (uint32_t i = 0; i < nb; ++i) {
        const int li = int(i & 63);
        const bool hi_half = i >= 64;
        const float wi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi_half ? w1 : w0), li));
        const int mi = __builtin_amdgcn_readlane(hi_half ? m1 : m0, li);
        const float d = U1[i * GB + i];
        const float qh = hs * (fminf(fmaxf(rintf(wi / hs) + hz, 0.f), maxq) - hz);          // high_quant.py:6-8
        const float t = wi - mu;
        const float sg = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
        const float ql = unfused(sg * sc) + mu;                                              // low_quant.py:75-82
        const float q = mi ? ql : qh;
        const float diff = wi - q;
        const float err = diff / d;
        loss += unfused(diff * diff) / unfused(d * d);                                       // gptq.py:155
        if (feedback) {                                                                      // gptq.py:159
            const float u0 = U1[i * GB + lane], u1 = U1[i * GB + 64 + lane];
            if (!hi_half && lane >= li) w0 -= unfused(err * u0);
            if (!hi_half || lane >= li) w1 -= unfused(err * u1);
        }
        if (lane == li) {
            if (hi_half) { q1 = q; e1 = err; } else { q0 = q; e0 = err; }
        }
    }
    if (in0) { wrow[lane] = q0; Err[size_t(r) * GB + lane] = e0; }
    if (in1) { wrow[lane + 64] = q1; Err[size_t(r) * GB + lane + 64] = e1; }
    if (!in0) Err[size_t(r) * GB + lane] = 0.f;
    if (!in1) Err[size_t(r) * GB + lane + 64] = 0.f;
    if (lane == 0) losses[r] += loss * 0.5f;                                                 // gptq.py:164
}

// HighQuantizer.calibrate(weight=True), per channel, asymmetric, no mse search (gptq_pb/high_quant.py:29-67,95-102):
// xmin = min(row, 0), xmax = max(row, 0) (both 0 -> -1/+1), scale = (xmax - xmin) / maxq, zero = round(-xmin / scale).
// One workgroup per row; correctly rounded divisions (torch's GPU division is not, and the codes depend on it).
__global__ __launch_bounds__(PT) void high_calibrate_kernel(const float* __restrict__ W, uint32_t K, float maxq,
                                                             float* __restrict__ scale, float* __restrict__ zero) {
    __shared__ float smin[PT], smax[PT];
    const float* wr = W + size_t(blockIdx.x) * K;
    float mn = 0.f, mx = 0.f;
    for (uint32_t j = threadIdx.x; j < K; j += PT) { const float v = wr[j]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = PT / 2; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float xmin = smin[0], xmax = smax[0];
        if (xmin == 0.f && xmax == 0.f) { xmin = -1.f; xmax = 1.f; }
        const float sc = (xmax - xmin) / maxq;
        scale[blockIdx.x] = sc;
        zero[blockIdx.x] = rintf(-xmin / sc);
    }
}

inline int launch(const void* k, int grid, int lds, void** argv, void* stream) {
    return hipLaunchKernel(k, dim3(grid), dim3(PT), argv, size_t(lds), static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK
                                                                                                                          : PBL_ERR_LAUNCH;
}

inline int grid_for(size_t n) {
    const size_t b = (n + PT - 1) / PT;
    return int(b < 4096 ? (b ? b : 1) : 4096);
}

}  // namespace

extern "C" {

size_t pbl_prep_workspace_bytes(void) { return sizeof(SelState) + size_t(2) * NBINS * 4; }

int pbl_kth_pair(const void* W, int w_dtype, size_t n, uint64_t k_lo, uint64_t k_hi, void* workspace, float* out2, void* stream) {
    if (!W || !workspace || !out2 || !n) return PBL_ERR_INVALID_ARG;
    if (k_lo < 1 || k_hi < 1 || k_lo > n || k_hi > n || n > 0xFFFFFFFFull) return PBL_ERR_INVALID_ARG;   // torch.kthvalue raises too
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return PBL_ERR_MISALIGNED;
    SelState* st = static_cast<SelState*>(workspace);
    uint32_t* hist = reinterpret_cast<uint32_t*>(st + 1);
    uint32_t kl = uint32_t(k_lo), kh = uint32_t(k_hi);
    const void* kh_fn = w_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(sel_hist<float>)
                      : w_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(sel_hist<_Float16>)
                      : w_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(sel_hist<__hip_bfloat16>) : nullptr;
    if (!kh_fn) return PBL_ERR_UNSUPPORTED;
    void* a0[] = {&st, &hist, &kl, &kh};
    int rc = launch(reinterpret_cast<const void*>(sel_init), 1, 0, a0, stream);
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    for (int p = 0; p < 3 && rc == PBL_OK; ++p) {
        int shift = shifts[p], bits = nbits[p], first = p == 0, last = p == 2;
        void* a1[] = {&W, &n, &st, &hist, &shift, &bits, &first};
        rc = launch(kh_fn, grid_for(n), 0, a1, stream);
        if (rc != PBL_OK) break;
        void* a2[] = {&st, &hist, &bits, &first, &last, &out2};
        rc = launch(reinterpret_cast<const void*>(sel_scan), 1, 0, a2, stream);
    }
    return rc;
}

int pbl_outlier_mask(const void* W, int w_dtype, size_t n, const float* thr2, uint8_t* mask_out, void* stream) {
    if (!W || !thr2 || !mask_out || !n) return PBL_ERR_INVALID_ARG;
    const void* k = w_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(mask_kernel<float>)
                  : w_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(mask_kernel<_Float16>)
                  : w_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(mask_kernel<__hip_bfloat16>) : nullptr;
    if (!k) return PBL_ERR_UNSUPPORTED;
    void* argv[] = {&W, &n, &thr2, &mask_out};
    return launch(k, grid_for(n), 0, argv, stream);
}

int pbl_quant8_rows(void* W, int w_dtype, uint32_t N, uint32_t K, float* code_scale, float* code_zp, void* stream) {
    if (!W || !code_scale || !code_zp || !N || !K) return PBL_ERR_INVALID_ARG;
    if (size_t(K) * 4 > 64 * 1024) return PBL_ERR_UNSUPPORTED;
    const void* k = w_dtype == PBL_DTYPE_F32 ? reinterpret_cast<const void*>(quant8_rows_kernel<float>)
                  : w_dtype == PBL_DTYPE_F16 ? reinterpret_cast<const void*>(quant8_rows_kernel<_Float16>)
                  : w_dtype == PBL_DTYPE_BF16 ? reinterpret_cast<const void*>(quant8_rows_kernel<__hip_bfloat16>) : nullptr;
    if (!k) return PBL_ERR_UNSUPPORTED;
    void* argv[] = {&W, &K, &code_scale, &code_zp};
    return launch(k, int(N), int(K) * 4, argv, stream);
}

int pbl_high_calibrate(const float* W, uint32_t N, uint32_t K, float maxq, float* scale, float* zero, void* stream) {
    if (!W || !scale || !zero || !N || !K || !(maxq > 0.f)) return PBL_ERR_INVALID_ARG;
    void* argv[] = {&W, &K, &maxq, &scale, &zero};
    return launch(reinterpret_cast<const void*>(high_calibrate_kernel), int(N), 0, argv, stream);
}

int pbl_gptq_block(float* W, uint32_t N, uint32_t K, uint32_t c0, uint32_t ncols, const float* U, const uint8_t* low_mask,
                   const float* hscale, const float* hzero, float maxq, const float* mean, const float* scale,
                   float* err_out, float* losses, int feedback, void* stream) {
    if (!W || !U || !low_mask || !hscale || !hzero || !mean || !scale || !err_out || !losses) return PBL_ERR_INVALID_ARG;
    if (!N || !ncols || ncols > GB || c0 + ncols > K) return PBL_ERR_INVALID_ARG;
    void* argv[] = {&W, &N, &K, &c0, &ncols, &U, &low_mask, &hscale, &hzero, &maxq, &mean, &scale, &err_out, &losses, &feedback};
    return launch(reinterpret_cast<const void*>(gptq_block_kernel), int((N + PT / 64 - 1) / (PT / 64)), GB * GB * 4, argv, stream);
}

}  // extern "C"
