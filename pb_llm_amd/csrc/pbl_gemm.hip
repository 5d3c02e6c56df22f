// pbl_gemm.hip -- small-batch GEMM (12 <= M <= 64) straight from the PBL1 packed format.
//
// For fp16-exact layers (PBL_FLAG_SAL_F16: packed from an fp16 checkpoint, every weight is an
// fp16 number) the GEMV's "stream the weights once per 4 tokens" costs M/4 weight passes.  Here
// one workgroup (4 waves) owns one record (16 output rows) and walks the K dimension in bands
// of 1024 columns:
//   A  expand the band's two sign-plane tiles to fp16 (hi | lo) in LDS            [16 x 1024]
//   B  scatter the record's salient code entries / exceptions that fall in the band (exact
//      fp16 values, double-rounded like the checkpoint)
//   C  v_mfma_f32_16x16x32_f16: A operand = band rows from LDS (ds_read_b128, rows padded
//      against bank conflicts), B operand = x[token][cols] fragments read from L2; each wave
//      takes every 4th k-step; fp32 accumulators for up to 4 token blocks of 16
// and finally reduces the 4 waves' accumulators through LDS and stores y (fp16, + bias).
// One pass over the packed weights for all M tokens; x (M*K*2 B) is re-read per record from
// L2, which bounds the useful M (dense unpack + library GEMM takes over above 64).
// Replaces F.linear(x, W_fq, b) of the reference for small prefill / batched decode
// (gptq_pb/eval_ppl_utils.py:59-60, BASELINE config 4).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GW 64
#define BAND 1024
#define BSTRIDE (BAND + 8)   // halves per LDS row: +16 B shifts consecutive rows by 4 banks

namespace {

__device__ __forceinline__ _Float16 round_f16_twice(float prod) {
    asm volatile("" : "+v"(prod));   // keep the fp32 product: the checkpoint value is double rounded
    return _Float16(prod);
}

template <int NTB>
__global__ __launch_bounds__(4 * GW) void pbl_gemm_band_kernel(pbl_layer L, const _Float16* __restrict__ x,
                                                                 _Float16* __restrict__ y, int M) {
    __shared__ __attribute__((aligned(16))) _Float16 band[16 * BSTRIDE];
    __shared__ __attribute__((aligned(16))) float red[4][NTB][GW][4];
    extern __shared__ __attribute__((aligned(16))) uint32_t span[];   // per chunk: first col | last col << 16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t rb = blockIdx.x;
    const int K = int(L.K), P = int(L.P), G = int(L.G);
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    const uint8_t* rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    const int nfull = __builtin_amdgcn_readfirstlane(info.y), ntail = __builtin_amdgcn_readfirstlane(info.z);
    const int nexc = __builtin_amdgcn_readfirstlane(info.w), nch = nfull + ntail;
    const bool groups = L.flags & PBL_FLAG_HAS_GROUPS;
    const uint32_t tiles_off = PBL_TILES_OFF(uint32_t(G));
    const uint32_t off_sal = tiles_off + uint32_t(P) * 1024u;
    const pbl_rowparams* params = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
    const pbl_rowinfo* rinfo = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF);
    const float2* ghl = reinterpret_cast<const float2*>(rec + PBL_REC_GHL_OFF);
    const int gwords = groups ? (K / G) / 128 : (1 << 30);
    const u32x4* tiles = reinterpret_cast<const u32x4*>(rec + tiles_off) + lane;
    const uint8_t* sal = rec + off_sal;
    const uint32_t nchu = uint32_t(nch);
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const u32x4* codep = reinterpret_cast<const u32x4*>(sal + PBL_SAL_CODE_OFF(nchu));
    const uint8_t* tailcnt = sal + PBL_SAL_TAILCNT_OFF(nchu);
    const bool has_crow = L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16);
    const uint8_t* crow = sal + PBL_SAL_CROW_OFF(nchu, uint32_t(ntail));
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));

    v4f acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};

    // column span of every salient chunk, once: a band pass then only decodes the chunks it intersects
    for (int c = tid; c < nch; c += 4 * GW) {
        const int cnt = c >= nfull ? int(tailcnt[c - nfull]) : 16;
        const u32x4 d4 = deltap[c];
        uint32_t last = col0p[c], first = last;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (e < cnt) last += ((d4[e >> 2] >> (8 * (e & 3))) & 0xFFu) >> 1;
        }
        span[c] = first | (last << 16);
    }
    const int row_a = lane & 15, kblk = lane >> 4;

    const int nbands = (K + BAND - 1) / BAND;
    for (int b = 0; b < nbands; ++b) {
        const int band0 = b * BAND;
        // x fragments of this wave's k-steps: issued first, they land while the band is expanded
        v8h bfr[BAND / 32 / 4][NTB];
#pragma unroll
        for (int kk = 0; kk < BAND / 32 / 4; ++kk) {
            const int kcol = band0 + (wave + 4 * kk) * 32 + kblk * 8;
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const int tok = t * 16 + row_a;
                v8h f = {0, 0, 0, 0, 0, 0, 0, 0};
                if (tok < M && kcol < K) {
                    const _Float16* src = x + size_t(tok) * K + kcol;
                    if (kcol + 8 <= K && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) f = *reinterpret_cast<const v8h*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = kcol + e < K ? src[e] : _Float16(0);
                    }
                }
                bfr[kk][t] = f;
            }
        }
        // ---- A: sign plane of panels 2b, 2b+1 -> fp16 (hi | lo); wave w does dword jobs w, w+4
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int job = wave + 4 * jj, p = 2 * b + (job >> 2), i = job & 3;
            uint32_t w = 0;
            if (p < P) w = reinterpret_cast<const uint32_t*>(tiles + p * 64)[i];
            const int g = (p * 4 + i) / gwords;
#pragma unroll
            for (int rho = 0; rho < 16; ++rho) {
                const int pos = rho < 8 ? rho + 8 : rho - 8;
                float hi, lo;
                if (groups) { const float2 hl = ghl[rho * G + (g < G ? g : G - 1)]; hi = hl.x; lo = hl.y; }
                else { hi = params[rho].hi; lo = params[rho].lo; }
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                h2 v;
                v.x = _Float16(((w >> pos) & 1u) ? hi : lo);
                v.y = _Float16(((w >> (16 + pos)) & 1u) ? hi : lo);
                *reinterpret_cast<h2*>(&band[rho * BSTRIDE + (job >> 2) * 512 + i * 128 + 2 * lane]) = v;
            }
        }
        __syncthreads();
        // ---- B: salient code entries and exceptions that fall into [band0, band0 + BAND)
        for (int c = tid; c < nch; c += 4 * GW) {
            const uint32_t sp = span[c];
            if (int(sp >> 16) < band0 || int(sp & 0xFFFFu) >= band0 + BAND) continue;
            int rho;
            if (has_crow) rho = crow[c];
            else {
                rho = 0;
                if (c < nfull) { for (int r = 1; r < 16; ++r) rho += int(rinfo[r].start) <= c; }
                else { const int t_ = c - nfull; for (int r = 1; r < 16; ++r) rho += int(rinfo[r].tailidx) <= t_; }
            }
            const int cnt = c >= nfull ? int(tailcnt[c - nfull]) : 16;
            const u32x4 d4 = deltap[c], q4 = codep[c];
            const float ss = params[rho].sscale, sz = params[rho].szero;
            int col = int(col0p[c]) - band0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                col += int(((d4[e >> 2] >> (8 * (e & 3))) & 0xFFu) >> 1);
                const _Float16 wv = round_f16_twice(ss * (float((q4[e >> 2] >> (8 * (e & 3))) & 0xFFu) - sz));
                if (e < cnt && col >= 0 && col < BAND) band[rho * BSTRIDE + col] = wv;
            }
        }
        for (int k = tid; k < nexc; k += 4 * GW) {
            const uint2 ex = exc[k];
            const int col = int(ex.x & 0xFFFFu) - band0;
            if (col >= 0 && col < BAND) band[int(ex.x >> 16) * BSTRIDE + col] = _Float16(__builtin_bit_cast(float, ex.y));
        }
        __syncthreads();
        // ---- C: MFMA over the band; wave w takes k-steps w, w+4, ...
#pragma unroll
        for (int kk = 0; kk < BAND / 32 / 4; ++kk) {
            const int ks = wave + 4 * kk;
            if (band0 + ks * 32 < K) {
                const v8h a = *reinterpret_cast<const v8h*>(&band[row_a * BSTRIDE + ks * 32 + kblk * 8]);
#pragma unroll
                for (int t = 0; t < NTB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfr[kk][t], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();   // the next band overwrites `band`
    }

    // ---- reduce the 4 waves' accumulators, add bias, store y[token][row] (4 rows = 8 bytes per lane)
#pragma unroll
    for (int t = 0; t < NTB; ++t) *reinterpret_cast<v4f*>(&red[wave][t][lane][0]) = acc[t];
    __syncthreads();
    if (wave != 0) return;
    const int tokl = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        v4f s = *reinterpret_cast<const v4f*>(&red[0][t][lane][0]);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const v4f*>(&red[w][t][lane][0]);
        const int tok = t * 16 + tokl;
        if (tok >= M) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t row = rb * 16 + r0 + r;
            if (row >= L.N) continue;
            float v = s[r];
            if (L.bias) v += L.bias[row];
            y[size_t(tok) * L.N + row] = _Float16(v);
        }
    }
}

}  // namespace

extern "C" int pbl_gemm_small_f16(const pbl_layer* layer, const void* x, void* y, int M, void* stream) {
    if (!layer || !layer->blob || !x || !y || M < 1 || M > 64) return PBL_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(layer->blob) & 15) return PBL_ERR_MISALIGNED;
    if (!(layer->flags & PBL_FLAG_SAL_F16)) return PBL_ERR_UNSUPPORTED;   // weights must be fp16-exact
    pbl_layer L = *layer;
    const _Float16* xp = static_cast<const _Float16*>(x);
    _Float16* yp = static_cast<_Float16*>(y);
    void* argv[] = {&L, &xp, &yp, &M};
    const int ntb = (M + 15) / 16;
    const void* k = ntb == 1   ? reinterpret_cast<const void*>(pbl_gemm_band_kernel<1>)
                    : ntb == 2 ? reinterpret_cast<const void*>(pbl_gemm_band_kernel<2>)
                    : ntb == 3 ? reinterpret_cast<const void*>(pbl_gemm_band_kernel<3>)
                               : reinterpret_cast<const void*>(pbl_gemm_band_kernel<4>);
    const size_t dyn = (size_t(layer->max_nch) * 4 + 15) & ~size_t(15);
    if (dyn > 12 * 1024) return PBL_ERR_UNSUPPORTED;   // static 49 KiB + spans must stay under 64 KiB
    return hipLaunchKernel(k, dim3(layer->NRB), dim3(4 * GW), argv, dyn, static_cast<hipStream_t>(stream)) == hipSuccess
               ? PBL_OK : PBL_ERR_LAUNCH;
}
