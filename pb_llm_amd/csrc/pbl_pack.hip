// pbl_pack.hip -- device-side packer: dense simulated weight (fp32, in HBM) -> PBL1 blob (in HBM), BYTE-IDENTICAL to the host
// packer pbl_pack_dense_f32 (pbl_host.cpp).  The reference has no packed format (gptq_pb/gptq.py:180-184 writes dense fp16
// back); round 1 packed on the host, so the QAT eval path (quant/outlier_quantizer.py:83-99 re-simulated weights) and
// LowHighGPTQ.to_pb() went GPU -> host -> GPU.  Records (16 rows) are independent: one wavefront per record.
//
// Lanes 0..15 each WALK one row left to right (the salient list of a row is cut into chunks greedily in column order, so a
// row is inherently sequential); all 16 walk the same column at the same time, which turns the sign plane into two
// wave ballots per dword (16 rows x 2 columns).  Two passes over W with the same classification code:
//   count  per row: full chunks, tail chunks, exceptions, coded entries -> record size (the caller prefix-sums the sizes)
//   write  every chunk goes straight to its final slot (full chunks of row 0, row 1, ... then the tails), the sign plane
//          is assembled in LDS and copied out, the slab index is accumulated per (row, slab) in LDS.
// Arithmetic mirrors the host's exactly: correctly rounded division (__fdiv_rn), nearbyint = rintf, the dequantisation
// product kept un-fused, the fp16 round trip through the fp32 product.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

namespace {

constexpr int CNT_STRIDE = PBL_PACK_COUNT_WORDS;     // u32 words per record in the count buffer

struct PackDevArgs {
    const float *W, *hi, *lo, *ss, *sz;
    const uint8_t* mask;
    uint32_t N, K, G, gs, P, NRB, flags;
    uint32_t* counts;            // [NRB][CNT_STRIDE]: {rec_bytes, nfull, ntail, nexc, nnz, 0,0,0, then 16 x {nfull_r, ntail_r, nexc_r}}
    const uint64_t* rec_off;     // write pass: byte offset of every record (+ end) from the blob start
    uint8_t* blob;
};

__device__ __forceinline__ float dequant_dev(float ss, float sz, int q, bool f16) {
    float p = __fmul_rn(ss, __fsub_rn(float(q), sz));
    asm volatile("" : "+v"(p));                      // keep the fp32 product (no fused multiply-convert)
    return f16 ? float(_Float16(p)) : p;
}

// the host packer's sal16_storable (pbl_host.cpp): fl16((-ss) * (q - sz)) must have a bit set
__device__ __forceinline__ bool sal16_storable_dev(float ss, float sz, int q) {
    float p = __fmul_rn(-ss, __fsub_rn(float(q), sz));
    asm volatile("" : "+v"(p));
    return __builtin_bit_cast(uint16_t, _Float16(p)) != 0;
}

template <bool WRITE>
__global__ __launch_bounds__(64) void pack_dev_kernel(PackDevArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x, N = a.N, K = a.K, G = a.G, P = a.P;
    const int rho = lane & 15;
    const uint32_t r = b * 16 + rho;
    const bool walker = lane < 16 && r < N;
    const bool sal16 = a.flags & PBL_FLAG_SAL_F16, has_crow = G > 1 || sal16;
    const uint32_t NS = PBL_NSLABS(K);
    // LDS (write pass): sign-plane image [P][256] dwords; per row: chunk buffer 16 steps + 16 codes; slab counters
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem_p);
    uint8_t* cbuf = reinterpret_cast<uint8_t*>(smem_p + (WRITE ? size_t(P) * 1024 : 0));       // [16 rows][32 B]
    uint16_t* fstart = reinterpret_cast<uint16_t*>(cbuf + 16 * 32);                             // [16][NS] full chunks starting in the slab
    uint16_t* tstart = fstart + 16 * NS;                                                        // [16][NS] tail chunks
    uint8_t* backs = reinterpret_cast<uint8_t*>(tstart + 16 * NS);                              // [16][NS] bit 0 fback, bit 1 tback
    if (WRITE) {
        for (uint32_t i = lane; i < P * 256; i += 64) tile[i] = 0;
        for (uint32_t i = lane; i < 16 * NS; i += 64) { fstart[i] = 0; tstart[i] = 0; backs[i] = 0; }
    }
    __syncthreads();

    // per-row constants
    const float ss = (walker && a.ss) ? a.ss[r] : 0.f, sz = (walker && a.sz) ? a.sz[r] : 0.f;
    float h = 0.f, l = 0.f;
    // final slots (write pass): this row's full chunks start at fstart_r, its tails at nfull_rec + tstart_r
    uint32_t nfull_rec = 0, ntail_rec = 0, nexc_rec = 0, start_r = 0, tailidx_r = 0, excoff_r = 0;
    uint8_t* rec = nullptr;
    uint16_t* col0p = nullptr; uint8_t *deltap = nullptr, *codep = nullptr, *tailcnt = nullptr, *crow = nullptr;
    pbl_exception* excp = nullptr;
    uint32_t* slabp = nullptr;
    const size_t fixed = size_t(PBL_TILES_OFF(G)) + size_t(P) * 1024;
    if (WRITE) {
        const uint32_t* cn = a.counts + size_t(b) * CNT_STRIDE;
        nfull_rec = cn[1]; ntail_rec = cn[2]; nexc_rec = cn[3];
        uint32_t nf = lane < 16 ? cn[8 + 3 * rho] : 0, nt = lane < 16 ? cn[8 + 3 * rho + 1] : 0, ne = lane < 16 ? cn[8 + 3 * rho + 2] : 0;
        // exclusive prefix over the 16 rows
        uint32_t pf = nf, pt = nt, pe = ne;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t of = __shfl_up(pf, d, 64), ot = __shfl_up(pt, d, 64), oe = __shfl_up(pe, d, 64);
            if (rho >= d) { pf += of; pt += ot; pe += oe; }
        }
        start_r = pf - nf; tailidx_r = pt - nt; excoff_r = pe - ne;
        rec = a.blob + a.rec_off[b];
        const uint32_t nch = nfull_rec + ntail_rec;
        uint8_t* s = rec + fixed;
        col0p = reinterpret_cast<uint16_t*>(s);
        deltap = s + PBL_SAL_DELTA_OFF(nch);
        codep = s + PBL_SAL_CODE_OFF(nch);
        tailcnt = s + PBL_SAL_TAILCNT_OFF(nch);
        crow = s + PBL_SAL_CROW_OFF(nch, ntail_rec);
        excp = reinterpret_cast<pbl_exception*>(s + PBL_SAL_EXC_OFF(nch, ntail_rec, has_crow));
        slabp = reinterpret_cast<uint32_t*>(s + PBL_SAL_SLAB_OFF(nch, ntail_rec, nexc_rec, has_crow));
        if (lane < 16) {
            // record header, rowinfo, params, group levels
            if (lane == 0) {
                pbl_rec_header hd = {nfull_rec, ntail_rec, nexc_rec, uint32_t(fixed)};
                *reinterpret_cast<pbl_rec_header*>(rec) = hd;
                pbl_rec_info ri = {uint32_t(a.rec_off[b] / 16), nfull_rec, ntail_rec, nexc_rec};
                reinterpret_cast<pbl_rec_info*>(a.blob + sizeof(pbl_blob_header))[b] = ri;
            }
            pbl_rowinfo rw;
            rw.start = uint16_t(start_r); rw.nfull = uint16_t(nf); rw.tailidx = uint16_t(tailidx_r); rw.ntail = uint8_t(nt); rw.pad = 0;
            reinterpret_cast<pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF)[rho] = rw;
            pbl_rowparams pr = {0.f, 0.f, 0.f, 0.f};
            if (r < N) { pr.hi = a.hi[size_t(r) * G]; pr.lo = a.lo[size_t(r) * G]; pr.sscale = ss; pr.szero = sz; }
            reinterpret_cast<pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF)[rho] = pr;
            if (G > 1 && r < N) {
                float* ghl = reinterpret_cast<float*>(rec + PBL_REC_GHL_OFF);
                for (uint32_t g = 0; g < G; ++g) { ghl[(size_t(rho) * G + g) * 2] = a.hi[size_t(r) * G + g]; ghl[(size_t(rho) * G + g) * 2 + 1] = a.lo[size_t(r) * G + g]; }
            }
        }
    }

    // ---- the walk -----------------------------------------------------------------------------------------------------
    uint32_t nf_r = 0, nt_r = 0, ne_r = 0, nnz_r = 0;     // counters (count pass) / cursors (write pass)
    bool open = false;
    uint32_t cnt = 0, first = 0, last = 0;
    uint8_t* mybuf = cbuf + rho * 32;
    auto close_chunk = [&]() {
        const bool full = cnt == 16;
        if (WRITE) {
            const uint32_t idx = full ? start_r + nf_r : nfull_rec + tailidx_r + nt_r;
            col0p[idx] = uint16_t(first);
            for (uint32_t k = cnt; k < 16; ++k) { mybuf[k] = 0; mybuf[16 + k] = mybuf[16 + cnt - 1]; }   // PBL_FLAG_TAIL_REPEAT padding
            const uint4 dv = *reinterpret_cast<const uint4*>(mybuf), qv = *reinterpret_cast<const uint4*>(mybuf + 16);
            *reinterpret_cast<uint4*>(deltap + size_t(idx) * 16) = dv;
            *reinterpret_cast<uint4*>(codep + size_t(idx) * 16) = qv;
            if (!full) tailcnt[tailidx_r + nt_r] = uint8_t(cnt);
            if (has_crow) crow[idx] = uint8_t(rho);
            const uint32_t s0 = first / PBL_SLAB_COLS, s1 = last / PBL_SLAB_COLS;
            (full ? fstart : tstart)[rho * NS + s0] += 1;
            for (uint32_t s = s0 + 1; s <= s1; ++s) backs[rho * NS + s] |= full ? 1 : 2;
        }
        if (full) ++nf_r; else ++nt_r;
        open = false;
    };
    uint32_t dw_acc = 0;
    for (uint32_t c = 0; c < K; ++c) {
        int bit = 0;
        if (walker) {
            if (c % a.gs == 0) { const uint32_t g = c / a.gs; h = a.hi[size_t(r) * G + g]; l = a.lo[size_t(r) * G + g]; }
            const float v = a.W[size_t(r) * K + c];
            const bool forced = a.mask && a.mask[size_t(r) * K + c];
            if (!forced && v == h) bit = 1;
            else if (!forced && v == l) bit = 0;
            else {
                bit = 1;                                   // sparse entries correct against `hi`
                int code = -1;
                if (a.ss && ss != 0.f && isfinite(v)) {
                    const float qf = rintf(__fadd_rn(__fdiv_rn(v, ss), sz));
                    const int q0 = int(qf);
#pragma unroll
                    for (int dq = 0; dq <= 2 && code < 0; ++dq) {
                        const int q = q0 + (dq == 0 ? 0 : (dq == 1 ? 1 : -1));
                        if (q >= 0 && q <= 255 && dequant_dev(ss, sz, q, sal16) == v && (!sal16 || sal16_storable_dev(ss, sz, q))) code = q;
                    }
                }
                if (code >= 0) {
                    if (open && (cnt == 16 || c - last > PBL_MAX_GAP)) close_chunk();
                    uint32_t d = 0;
                    if (!open) { open = true; cnt = 0; first = c; } else d = 2 * (c - last);
                    if (WRITE) { mybuf[cnt] = uint8_t(d); mybuf[16 + cnt] = uint8_t(code); }
                    ++cnt; last = c; ++nnz_r;
                } else {
                    if (WRITE) { pbl_exception e = {uint16_t(c), uint16_t(rho), v}; excp[excoff_r + ne_r] = e; }
                    ++ne_r;
                }
            }
        }
        if (WRITE) {
            // 16 rows' bits of this column -> half of a sign-plane dword (bit pos(rho) + 16 e): rows 0..7 sit on 8..15, rows 8..15 on 0..7
            const uint32_t m = uint32_t(__ballot(bit != 0)) & 0xFFFFu;
            const uint32_t half = ((m & 0xFFu) << 8) | (m >> 8);
            dw_acc |= half << (16 * (c & 1));
            if ((c & 1) || c + 1 == K) {
                if (lane == 0) tile[(c / 512) * 256 + ((c % 128) / 2) * 4 + (c % 512) / 128] = dw_acc;
                dw_acc = 0;
            }
        }
    }
    if (walker && open) close_chunk();

    if (!WRITE) {
        // row counts -> record totals and size
        uint32_t tf = lane < 16 ? nf_r : 0, tt = lane < 16 ? nt_r : 0, te = lane < 16 ? ne_r : 0, tn = lane < 16 ? nnz_r : 0;
        uint32_t* cn = a.counts + size_t(b) * CNT_STRIDE;
        if (lane < 16) { cn[8 + 3 * rho] = tf; cn[8 + 3 * rho + 1] = tt; cn[8 + 3 * rho + 2] = te; }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { tf += __shfl_xor(tf, d, 64); tt += __shfl_xor(tt, d, 64); te += __shfl_xor(te, d, 64); tn += __shfl_xor(tn, d, 64); }
        if (lane == 0) {
            const uint32_t nch = tf + tt;
            cn[0] = uint32_t(fixed + PBL_SAL_BYTES(nch, tt, te, has_crow, K));
            cn[1] = tf; cn[2] = tt; cn[3] = te; cn[4] = tn;
            // status word: limits of the format (u16 chunk indices, u8 tail count per row)
            cn[5] = (nch > 65535u) ? 1u : 0u;
            cn[6] = 0; cn[7] = 0;
        }
        if (lane < 16 && nt_r > 255) atomicOr(&cn[5], 1u);
        return;
    }
    __syncthreads();
    // sign plane image -> record
    {
        const uint4* src = reinterpret_cast<const uint4*>(tile);
        uint4* dst = reinterpret_cast<uint4*>(rec + PBL_TILES_OFF(G));
        for (uint32_t i = lane; i < P * 64; i += 64) dst[i] = src[i];
    }
    // slab index: cumulative chunk counts per (row, slab)
    if (lane < 16) {
        uint32_t fe = 0, te = 0;
        for (uint32_t s = 0; s < NS; ++s) {
            fe += fstart[rho * NS + s]; te += tstart[rho * NS + s];
            const uint32_t bk = backs[rho * NS + s];
            slabp[rho * NS + s] = fe | (te << 16) | ((bk & 1u) << 24) | (((bk >> 1) & 1u) << 25);
        }
    }
}

size_t pack_lds(uint32_t P, uint32_t NS, bool write) {
    return (write ? size_t(P) * 1024 : 0) + 16 * 32 + size_t(16) * NS * 5 + 64;
}

}  // namespace

extern "C" {

int pbl_pack_dev_count(const float* W, uint32_t N, uint32_t K, uint32_t G, const float* hi, const float* lo,
                       const float* sscale, const float* szero, const uint8_t* sal_mask, uint32_t flags,
                       uint32_t* counts_out, void* stream) {
    if (flags & ~PBL_FLAG_SAL_F16) return PBL_ERR_INVALID_ARG;
    if (!W || !hi || !lo || !counts_out || N == 0 || K == 0 || G == 0) return PBL_ERR_INVALID_ARG;
    if (K > 32767 || N > (1u << 24)) return PBL_ERR_UNSUPPORTED;
    if (G > 1 && (K % G != 0 || (K / G) % 128 != 0)) return PBL_ERR_UNSUPPORTED;
    PackDevArgs a{};
    a.W = W; a.hi = hi; a.lo = lo; a.ss = sscale; a.sz = szero; a.mask = sal_mask;
    a.N = N; a.K = K; a.G = G; a.gs = K / G; a.P = (K + PBL_PANEL_COLS - 1) / PBL_PANEL_COLS; a.NRB = (N + 15) / 16; a.flags = flags;
    a.counts = counts_out;
    void* argv[] = {&a};
    return hipLaunchKernel(reinterpret_cast<const void*>(pack_dev_kernel<false>), dim3(a.NRB), dim3(64), argv,
                           pack_lds(a.P, PBL_NSLABS(K), false), static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

int pbl_pack_dev_write(const float* W, uint32_t N, uint32_t K, uint32_t G, const float* hi, const float* lo,
                       const float* sscale, const float* szero, const uint8_t* sal_mask, uint32_t flags,
                       const uint32_t* counts, const uint64_t* rec_off, uint64_t blob_bytes, uint32_t max_nch, uint32_t max_nexc,
                       uint64_t nnz, uint64_t nexc, void* blob_out, void* stream) {
    if (flags & ~PBL_FLAG_SAL_F16) return PBL_ERR_INVALID_ARG;
    if (!W || !hi || !lo || !counts || !rec_off || !blob_out || N == 0 || K == 0 || G == 0) return PBL_ERR_INVALID_ARG;
    if (K > 32767 || N > (1u << 24)) return PBL_ERR_UNSUPPORTED;
    if (G > 1 && (K % G != 0 || (K / G) % 128 != 0)) return PBL_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(blob_out) & 15) return PBL_ERR_MISALIGNED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    PackDevArgs a{};
    a.W = W; a.hi = hi; a.lo = lo; a.ss = sscale; a.sz = szero; a.mask = sal_mask;
    a.N = N; a.K = K; a.G = G; a.gs = K / G; a.P = (K + PBL_PANEL_COLS - 1) / PBL_PANEL_COLS; a.NRB = (N + 15) / 16; a.flags = flags;
    a.counts = const_cast<uint32_t*>(counts); a.rec_off = rec_off; a.blob = static_cast<uint8_t*>(blob_out);
    const size_t lds = pack_lds(a.P, PBL_NSLABS(K), true);
    if (lds > 160 * 1024) return PBL_ERR_UNSUPPORTED;
    const void* k = reinterpret_cast<const void*>(pack_dev_kernel<true>);
    if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) return PBL_ERR_LAUNCH;
    if (hipMemsetAsync(blob_out, 0, blob_bytes, st) != hipSuccess) return PBL_ERR_LAUNCH;          // padding bytes are zero, like the host's
    pbl_blob_header h;
    __builtin_memset(&h, 0, sizeof(h));
    h.magic = PBL_MAGIC; h.version = PBL_VERSION; h.N = N; h.K = K; h.P = a.P; h.G = G; h.NRB = a.NRB;
    h.flags = (G > 1 ? PBL_FLAG_HAS_GROUPS : 0) | ((flags & PBL_FLAG_SAL_F16) ? PBL_FLAG_SAL_F16 : 0) | PBL_FLAG_TAIL_REPEAT | PBL_FLAG_SLABS;
    h.max_nch = max_nch; h.max_nexc = max_nexc; h.nnz = nnz; h.nexc = nexc; h.blob_bytes = blob_bytes; h.rb_off_pos = uint32_t(sizeof(pbl_blob_header));
    // header and the closing rb_info entry: two small host -> device copies on the stream (pageable source: copied before return)
    if (hipMemcpyAsync(blob_out, &h, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess) return PBL_ERR_LAUNCH;
    pbl_rec_info endinfo = {uint32_t(blob_bytes / 16), 0, 0, 0};
    if (hipMemcpyAsync(static_cast<uint8_t*>(blob_out) + sizeof(h) + size_t(a.NRB) * sizeof(pbl_rec_info), &endinfo, sizeof(endinfo),
                       hipMemcpyHostToDevice, st) != hipSuccess) return PBL_ERR_LAUNCH;
    void* argv[] = {&a};
    return hipLaunchKernel(k, dim3(a.NRB), dim3(64), argv, lds, st) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

}  // extern "C"
