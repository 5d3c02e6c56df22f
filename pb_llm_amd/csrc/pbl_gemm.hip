// pbl_gemm.hip -- small-batch GEMM (1 <= M <= 32 tokens) straight from the PBL1 packed format, on the
// matrix cores.  The GEMV streams the weights once per 4 tokens; this kernel streams them ONCE for up to
// 32 tokens (batched decode, short prefill: BASELINE config 4).  Replaces F.linear(x, W_fq, b) of the
// reference (quant/outlier_quantizer.py:105, gptq_pb/eval_ppl_utils.py:59-60).  Any layer with G == 1.
//
// One workgroup (2 waves when the layer has >= 512 records, else 4) owns one 16-row record; its panels are
// split over the waves and the partial accumulators are combined at the end in a fixed order.  The record's salient
// chunks are first counting-sorted by 256-column half panel (LDS, whole workgroup), so a half panel
// touches only its own chunks.  Per half panel, a wave
//   * copies its half of the 1 KiB sign-plane tile to LDS twice (Wp: as is, and shifted left 8 for rows 8..15);
//   * writes the salient entries into an fp16 tile St[16][256] (the uint8 code -- exact in fp16 -- or, for
//     fp16 checkpoints, the double-rounded fp16 weight) and a byte tile Mt[16][256] (0x3C = the high byte of
//     fp16 1.0 at salient positions).  The scatter is branch-free: a column outside the half panel is
//     clamped (v_min_u32) into the row's pad column, and tail-chunk padding repeats the last entry
//     (PBL_FLAG_TAIL_REPEAT), so all 16 entries of every chunk are simply written;
//   * per 32-column k-step reads 4 plane dwords (one b128, broadcast over the 16 row-lanes) and turns them
//     into a CLASS-CODED fp16 A fragment with one v_and_or per dword -- the GEMV's (w & M_c) | C_c trick, now
//     indexed by the lane's own row; reads 8 mask bytes and widens them to fp16 {0, 1} with four v_perm_b32;
//     and runs v_mfma_f32_16x16x32_f16 for  accW += W.x, accM += Mask.x, accS += St.x, accX += 1.x;
//   * clears Mt with wide stores (St is never cleared: stale entries are multiplied by the 0/1 mask when read).
// Decode in fp32:  D = A_c*accW - B_c*X (sum of +-1 * x),  S = accM,
//   y = alpha*D + mu*X + [ss*(Q - sz*S) | Q] - hi*S + exceptions + bias.
// One pass over the packed weights for up to 32 tokens; x fragments come from L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GW 64
#define PW 256
#define SSTR (PW + 8)
// performance-analysis hook (tools/ablate_mfma.sh): bit 0 no bucket sort, 1 no scatter, 2 no MFMA,
// 3 no clears, 4 no x loads, 6 scatter arithmetic without the LDS writes, 7 constant A fragments (no LDS reads),
// 9 stop after the sort.  0 in every shipped build.
#ifndef PBL_MFMA_ABLATE
#define PBL_MFMA_ABLATE 0
#endif

namespace {

__device__ __forceinline__ _Float16 round_f16_twice(float prod) {
    asm volatile("" : "+v"(prod));   // keep the fp32 product: the checkpoint value is double rounded
    return _Float16(prod);
}

__device__ __forceinline__ void class_consts_g(int ci, float& A, float& B, uint32_t& Cc) {
    const float a[8] = {8.f, 4.f, 32768.f, 16384.f, 4096.f, 256.f, 1.f, -1.f};
    const float b[8] = {9.f, 5.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f};
    A = a[0]; B = b[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { A = ci == k ? a[k] : A; B = ci == k ? b[k] : B; }
    Cc = (ci < 2 || ci == 7) ? 0x3C003C00u : 0u;   // classes whose two values are 1.0 + {0, d}
}

// one salient chunk held in registers: its 16 byte-steps, 16 codes, first column, row
struct ChunkRegs {
    u32x4 d4, q4;
    int col0, rho;          // col0 == PBL_NO_CHUNK: nothing to do (every column clamps into the pad)
};
#define PBL_NO_CHUNK (1 << 20)

constexpr size_t MFMA_WAVE_BYTES = size_t(16) * SSTR * 2 + size_t(16) * SSTR + 2 * 512;   // St + Mt + Wp + Wp << 8

// LDS bytes of the workgroup-shared part (after the 4 waves' tiles)
__host__ __device__ inline size_t mfma_shared_bytes(int NH, int list_cap, int max_nch) {
    return size_t(256) + 128 + ((size_t(max_nch) + 15) & ~size_t(15)) + size_t(2 * (NH + 1)) * 4 + ((size_t(list_cap) * 2 + 15) & ~size_t(15));
}

template <int NTB, bool SF, int WPG>
__global__ __launch_bounds__(WPG * GW) void pbl_mfma_kernel(pbl_layer L, const _Float16* __restrict__ x,
                                                            void* __restrict__ yv, int M, int y_f32, int list_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t rb = blockIdx.x;                // one 16-row record per workgroup
    const int K = int(L.K), P = int(L.P);
    const int NB = (K + 127) / 128;                // 128-column sub-blocks
    const int NH = (K + PW - 1) / PW;              // 256-column half panels = sort buckets
    constexpr bool sf = SF;
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    const uint8_t* rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    const int nfull = __builtin_amdgcn_readfirstlane(info.y), ntail = __builtin_amdgcn_readfirstlane(info.z);
    const int nexc = __builtin_amdgcn_readfirstlane(info.w), nch = nfull + ntail;
    const uint32_t nchu = uint32_t(nch);
    const uint32_t tiles_off = PBL_TILES_OFF(1u);
    const uint8_t* sal = rec + tiles_off + uint32_t(P) * 1024u;
    const pbl_rowparams* params = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
    const pbl_rowinfo* rinfo = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF);
    const u32x4* tiles = reinterpret_cast<const u32x4*>(rec + tiles_off) + lane;
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const u32x4* codep = reinterpret_cast<const u32x4*>(sal + PBL_SAL_CODE_OFF(nchu));
    constexpr bool has_crow = SF;                  // G == 1 here, so per-chunk row ids exist exactly for fp16 checkpoints
    const uint8_t* crow = sal + PBL_SAL_CROW_OFF(nchu, uint32_t(ntail));
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));

    // LDS: per wave St[16][SSTR] fp16, Mt[16][SSTR] bytes, Wp[128] and Wp8[128] dwords (sign plane of the half
    // panel, [sub-block][lane], as is and << 8); shared: row params, bucket offsets / cursors, bucket lists.  The per-chunk panel span used by the
    // sort lives in the (not yet used) tile area.
    _Float16* St = reinterpret_cast<_Float16*>(smem_g + size_t(wave) * MFMA_WAVE_BYTES);
    uint8_t* Mt = reinterpret_cast<uint8_t*>(St + 16 * SSTR);
    uint32_t* Wp = reinterpret_cast<uint32_t*>(Mt + 16 * SSTR);
    uint32_t* Wp8 = Wp + 128;
    char* shared = smem_g + WPG * MFMA_WAVE_BYTES;
    float4* prm = reinterpret_cast<float4*>(shared);                        // [16] {hi, lo, sscale, szero}
    int* rinf = reinterpret_cast<int*>(shared + 256);                       // [16] first full chunk, [16] first tail chunk of each row
    uint8_t* crow_l = reinterpret_cast<uint8_t*>(shared + 384);             // [nch] row of each chunk
    uint32_t* bstart = reinterpret_cast<uint32_t*>(shared + 384 + ((size_t(L.max_nch) + 15) & ~size_t(15)));   // [NH + 1]
    uint32_t* bfill = bstart + (NH + 1);                                    // [NH + 1]
    uint16_t* blist = reinterpret_cast<uint16_t*>(bfill + (NH + 1));       // [list_cap]
    uint16_t* span = reinterpret_cast<uint16_t*>(smem_g);                   // [nch] first panel | last panel << 8

    // this wave's panels [p_lo, p_hi) -> sub-blocks [b_lo, b_hi); first tile / x loads go out before the sort
    const int Pq = (P + WPG - 1) / WPG;
    const int p_lo = min(wave * Pq, P), p_hi = min(p_lo + Pq, P);
    const int b_hi = min(4 * p_hi, NB);
    const int h_lo = 2 * p_lo, h_hi = min(2 * p_hi, NH);
    const int row_a = lane & 15, kblk = lane >> 4;
    auto load_x = [&](int cb, v8h (&bfr)[4][NTB]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kcol = cb + ks * 32 + kblk * 8;
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const int tok = t * 16 + row_a;
                v8h f = {0, 0, 0, 0, 0, 0, 0, 0};
                if (!(PBL_MFMA_ABLATE & 16) && tok < M && kcol < K) f = *reinterpret_cast<const v8h*>(x + size_t(tok) * K + kcol);
                bfr[ks][t] = f;
            }
        }
    };
    u32x4 t_cur = {0, 0, 0, 0};
    if (p_lo < p_hi) t_cur = __builtin_nontemporal_load(tiles + p_lo * 64);

    // ---- one-time, whole workgroup: counting-sort the salient chunks by panel ----
    for (int i = tid; i < 2 * (NH + 1); i += WPG * GW) bstart[i] = 0;
    if (tid < 16) {
        prm[tid] = reinterpret_cast<const float4*>(params)[tid];
        rinf[tid] = int(rinfo[tid].start);
        rinf[16 + tid] = int(rinfo[tid].tailidx);
    }
    __syncthreads();
#pragma unroll 2
    for (int c = tid; c < ((PBL_MFMA_ABLATE & 1) ? 0 : nch); c += WPG * GW) {      // pass 1: bucket sizes
        const u32x4 d4 = deltap[c];                 // padding steps of a tail chunk are 0
        uint32_t last = col0p[c];
        const uint32_t b0 = last / PW;
#pragma unroll
        for (int e = 0; e < 16; ++e) last += ((d4[e >> 2] >> (8 * (e & 3))) & 0xFFu) >> 1;
        const uint32_t b1 = min(last / PW, uint32_t(NH - 1));
        span[c] = uint16_t(b0 | (b1 << 8));
        int rho = 0;                                // the chunk's row: stored per chunk, or from the row table
        if (has_crow) rho = crow[c];
        else if (c < nfull) { for (int q = 1; q < 16; ++q) rho += rinf[q] <= c; }
        else { for (int q = 1; q < 16; ++q) rho += rinf[16 + q] <= c - nfull; }
        crow_l[c] = uint8_t(rho);
        for (uint32_t b = b0; b <= b1; ++b) atomicAdd(&bstart[b + 1], 1u);
    }
    __syncthreads();
    if (wave == 0) {                               // inclusive scan of P + 1 <= 65 counters by one wave
        const int per = (NH + 1 + GW - 1) / GW, lo_i = lane * per;
        uint32_t loc = 0;
        for (int q = 0; q < per; ++q) if (lo_i + q <= NH) loc += bstart[lo_i + q];
        uint32_t inc = loc;
#pragma unroll
        for (int d = 1; d < GW; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        uint32_t run = inc - loc;
        for (int q = 0; q < per; ++q) if (lo_i + q <= NH) { run += bstart[lo_i + q]; bstart[lo_i + q] = run; }
    }
    __syncthreads();
    const bool lists_ok = bstart[NH] <= uint32_t(list_cap);    // uniform; else every panel scans every chunk
    if (lists_ok) {
        for (int c = tid; c < ((PBL_MFMA_ABLATE & 1) ? 0 : nch); c += WPG * GW) {  // pass 2: fill (order inside a bucket is irrelevant)
            const uint32_t sp = span[c];
            for (uint32_t b = sp & 0xFFu; b <= (sp >> 8); ++b) blist[bstart[b] + atomicAdd(&bfill[b], 1u)] = uint16_t(c);
        }
    }
    __syncthreads();                                // span is dead from here: the tile area becomes tiles
    for (int i = lane; i < int(MFMA_WAVE_BYTES / 16); i += GW)
        reinterpret_cast<u32x4*>(St)[i] = u32x4{0, 0, 0, 0};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    auto load_chunk = [&](int c, bool valid) -> ChunkRegs {
        ChunkRegs r;
        r.col0 = PBL_NO_CHUNK; r.rho = lane & 15;   // idle lanes spread their (pad) writes over the rows
        r.d4 = u32x4{0, 0, 0, 0}; r.q4 = u32x4{0, 0, 0, 0};
        if (valid) {
            r.d4 = deltap[c]; r.q4 = codep[c]; r.col0 = int(col0p[c]);
            r.rho = crow_l[c];
        }
        return r;
    };
    // all 16 entries of one chunk -> St / Mt of the half panel starting at column cb; an entry outside the
    // half panel lands in one of the row's 8 pad columns (PW .. PW+7), which no fragment reads
    const uint32_t padcol = uint32_t(PW + (lane & 7));   // 8 pad columns x 16 rows: out-of-range writes do not pile up on one address
    auto scatter = [&](const ChunkRegs& r, int cb) {
        if (PBL_MFMA_ABLATE & 2) return;
        int col = r.col0 - cb;
        const float4 pr = prm[r.rho];
        _Float16* strow = St + r.rho * SSTR;
        uint8_t* mrow = Mt + r.rho * SSTR;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            col += int(((r.d4[e >> 2] >> (8 * (e & 3))) & 0xFFu) >> 1);
            const uint32_t cc = min(uint32_t(col), padcol);
            const float qf = float((r.q4[e >> 2] >> (8 * (e & 3))) & 0xFFu);
            const _Float16 val = sf ? round_f16_twice(pr.z * (qf - pr.w)) : _Float16(qf);
            if (PBL_MFMA_ABLATE & 64) { if (cc == 0xFFFFu) strow[0] = val; }
            else { strow[cc] = val; mrow[cc] = 0x3C; }
        }
    };

    v4f accW[NTB], accS[NTB], accM[NTB], accX[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) { accW[t] = accS[t] = accM[t] = accX[t] = v4f{0.f, 0.f, 0.f, 0.f}; }
    v8h ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = _Float16(1.0f);
    // the lane's row decides its bit (shift rows >= 8 up by 8, then bit 8 + class in each half-word) and class
    float Acl, Bcl;
    uint32_t Ccl;
    class_consts_g(row_a & 7, Acl, Bcl, Ccl);
    const uint32_t Mcl = 0x01000100u << (row_a & 7);
    const uint32_t* Wsel = row_a >= 8 ? Wp8 : Wp;   // rows 8..15 read the pre-shifted copy
    auto frag = [&](const u32x4 d) -> v8h {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (d[q] & Mcl) | Ccl;
        return __builtin_bit_cast(v8h, o);
    };
    auto mfrag = [&](const uint2 m) -> v8h {         // 8 bytes {0, 0x3C} -> 8 fp16 {0, 1}: byte -> high byte
        u32x4 o;
        o[0] = __builtin_amdgcn_perm(m.x, 0u, 0x050C040Cu); o[1] = __builtin_amdgcn_perm(m.x, 0u, 0x070C060Cu);
        o[2] = __builtin_amdgcn_perm(m.y, 0u, 0x050C040Cu); o[3] = __builtin_amdgcn_perm(m.y, 0u, 0x070C060Cu);
        return __builtin_bit_cast(v8h, o);
    };

    if (!(PBL_MFMA_ABLATE & 512) && h_lo < h_hi) {
        // software pipeline over half panels: half panel h+1's first 128 bucket chunks (and the next panel's tile)
        // are requested at the top of half panel h; its x fragments as soon as h's MFMAs have consumed theirs
        auto bucket_chunk = [&](int h, int k2) -> ChunkRegs {
            const uint32_t s0 = bstart[h] + uint32_t(k2 * GW + lane);
            const bool v = lists_ok && h < h_hi && s0 < bstart[h + 1];
            return load_chunk(v ? int(blist[v ? s0 : 0]) : 0, v);
        };
        struct HalfPanel { ChunkRegs ca, cb; };
        auto issue = [&](int h, HalfPanel& s) {
            s.ca = bucket_chunk(h, 0);
            s.cb = bucket_chunk(h, 1);
        };
        v8h bx[2][4][NTB];                          // x fragments of the current half panel (beyond K: zeros)
        load_x(2 * h_lo * 128, bx[0]);
        load_x((2 * h_lo + 1) * 128, bx[1]);
        u32x4 t_next = t_cur;
        auto body = [&](int h, const int half, HalfPanel& cur, HalfPanel& nxt) {
            const int pc = h * PW;
            const uint32_t bs = bstart[h], be = bstart[h + 1];
            if (half == 0 && (h >> 1) + 1 < p_hi) t_next = __builtin_nontemporal_load(tiles + ((h >> 1) + 1) * 64);
            issue(h + 1, nxt);
            // this half panel's two sign-plane dwords -> Wp[sub-block][lane], as is and << 8
            {
                const uint32_t w0 = half ? t_cur[2] : t_cur[0], w1 = half ? t_cur[3] : t_cur[1];
                Wp[lane] = w0; Wp[64 + lane] = w1; Wp8[lane] = w0 << 8; Wp8[64 + lane] = w1 << 8;
            }
            // salient entries of the half panel -> St / Mt
            if (be > bs) scatter(cur.ca, pc);
            if (be > bs + GW) scatter(cur.cb, pc);
            if (lists_ok) {
                for (uint32_t idx = bs + 2 * GW + lane; idx < be; idx += GW) scatter(load_chunk(int(blist[idx]), true), pc);
            } else {                                // bucket lists did not fit: scan every chunk (slow, correct)
                for (int c = lane; c < nch; c += GW) scatter(load_chunk(c, true), pc);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (2 * h + i < b_hi) {             // uniform
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const v8h aW = (PBL_MFMA_ABLATE & 128) ? ones : frag(*reinterpret_cast<const u32x4*>(Wsel + i * 64 + ks * 16 + kblk * 4));
                        const v8h aM = (PBL_MFMA_ABLATE & 128) ? ones : mfrag(*reinterpret_cast<const uint2*>(Mt + row_a * SSTR + i * 128 + ks * 32 + kblk * 8));
                        // St is never cleared: entries left over from earlier half panels are multiplied by the mask (0 / 1, exact)
                        const v8h aS = (PBL_MFMA_ABLATE & 128) ? ones : *reinterpret_cast<const v8h*>(St + row_a * SSTR + i * 128 + ks * 32 + kblk * 8) * aM;
#pragma unroll
                        for (int t = 0; t < ((PBL_MFMA_ABLATE & 4) ? 0 : NTB); ++t) {
                            accW[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aW, bx[i][ks][t], accW[t], 0, 0, 0);
                            accS[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aS, bx[i][ks][t], accS[t], 0, 0, 0);
                            accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aM, bx[i][ks][t], accM[t], 0, 0, 0);
                            accX[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, bx[i][ks][t], accX[t], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (h + 1 < h_hi) {                     // the x registers are free: next half panel's fragments, hidden by
                load_x((2 * h + 2) * 128, bx[0]);   // the clear and the next scatter
                load_x((2 * h + 3) * 128, bx[1]);
            }
            if (!(PBL_MFMA_ABLATE & 8) && be > bs) {     // clear the mask tile only (Wp is overwritten, St is masked when read)
                for (int q = lane; q < 16 * SSTR / 16; q += GW) reinterpret_cast<u32x4*>(Mt)[q] = u32x4{0, 0, 0, 0};
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (half) t_cur = t_next;
        };
        HalfPanel hpA, hpB;
        issue(h_lo, hpA);
        for (int h = h_lo; h < h_hi; h += 2) {      // h_lo is even: the first body of a pair is the panel's first half
            body(h, 0, hpA, hpB);
            if (h + 1 < h_hi) body(h + 1, 1, hpB, hpA);
        }
    }

    // ---- combine the WPG K-slices in a fixed order; wave w then decodes accumulator components r = w, w + WPG, .. ----
    __syncthreads();                                // tiles are dead: reuse them as the reduction buffer
    float* red = reinterpret_cast<float*>(smem_g);  // [wave][4 * NTB accumulators][4 components][64 lanes]
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // accumulator rows are 4*kblk + r: raw sums are exchanged, the class decode follows the combine
            red[((wave * 4 * NTB + 0 * NTB + t) * 4 + r) * GW + lane] = accW[t][r];
            red[((wave * 4 * NTB + 1 * NTB + t) * 4 + r) * GW + lane] = accS[t][r];
            red[((wave * 4 * NTB + 2 * NTB + t) * 4 + r) * GW + lane] = accM[t][r];
            red[((wave * 4 * NTB + 3 * NTB + t) * 4 + r) * GW + lane] = accX[t][r];
        }
    __syncthreads();
    for (int r = wave; r < 4; r += WPG) {           // lane holds token (lane & 15) of each block, row 4*(lane >> 4) + r
    const int rho = 4 * kblk + r;
    const uint32_t row = rb * 16 + rho;
    if (row >= L.N) continue;
    const float4 pr = prm[rho];                     // {hi, lo, sscale, szero}
    float A, B;
    uint32_t Cunused;
    class_consts_g(rho & 7, A, B, Cunused);
    const float alpha = 0.5f * (pr.x - pr.y), mu = 0.5f * (pr.x + pr.y);
    const float bias = L.bias ? L.bias[row] : 0.f;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        const int tok = t * 16 + row_a;
        if (tok >= M) continue;
        float sums[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float v = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < WPG; ++w4) v += red[((w4 * 4 * NTB + a * NTB + t) * 4 + r) * GW + lane];
            sums[a] = v;
        }
        const float Wv = sums[0], Q = sums[1], Mv = sums[2], X = sums[3];
        const float D = fmaf(A, Wv, -(B * X));
        const float S = Mv;
        const float salv = sf ? fmaf(-pr.x, S, Q) : fmaf(pr.z, fmaf(-pr.w, S, Q), -(pr.x * S));
        float e = 0.f;
        for (int k = 0; k < nexc; ++k) {
            const uint2 ex = exc[k];
            if (int(ex.x >> 16) == rho)
                e += (__builtin_bit_cast(float, ex.y) - pr.x) * float(x[size_t(tok) * K + (ex.x & 0xFFFFu)]);
        }
        const float out = fmaf(alpha, D, fmaf(mu, X, salv)) + e + bias;
        if (y_f32) static_cast<float*>(yv)[size_t(tok) * L.N + row] = out;
        else static_cast<_Float16*>(yv)[size_t(tok) * L.N + row] = _Float16(out);
    }
    }
}

}  // namespace

extern "C" int pbl_gemm_mfma_f16(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* stream) {
    if (!layer || !layer->blob || !x || !y || M < 1 || M > 32) return PBL_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(layer->blob) & 15) return PBL_ERR_MISALIGNED;
    if (layer->G != 1 || (layer->K & 7) || (reinterpret_cast<uintptr_t>(x) & 15)) return PBL_ERR_UNSUPPORTED;
    if (!(layer->flags & PBL_FLAG_TAIL_REPEAT)) return PBL_ERR_UNSUPPORTED;   // the scatter writes all 16 entries of a chunk
    // bucket lists: a chunk of 16 salient entries spans about 16 / density columns, i.e. it lands in about
    // 1 + K / (16 * chunks-per-record) half panels of 256 columns; size for that plus 15 % and let the kernel fall
    // back to scanning for a record that still overflows (pathological gaps: slow, correct)
    const int NH = (int(layer->K) + PW - 1) / PW;
    if (NH > 255) return PBL_ERR_UNSUPPORTED;
    int list_cap = int(1.15 * (double(layer->max_nch) + double(layer->K) / 16.0)) + 64;
    // many records: 2 waves per record (more independent workgroups per CU, one round over the chip);
    // few records: 4 waves per record (the K split is the only parallelism there is)
    const int wpg = layer->NRB >= 512 ? 2 : 4;
    const size_t tiles_bytes = size_t(wpg) * MFMA_WAVE_BYTES;
    if (size_t(layer->max_nch) * 2 > tiles_bytes) return PBL_ERR_UNSUPPORTED;   // sort scratch lives in the tile area
    while (list_cap > 64 && tiles_bytes + mfma_shared_bytes(NH, list_cap, int(layer->max_nch)) > 160 * 1024) list_cap /= 2;
    const size_t lds = tiles_bytes + mfma_shared_bytes(NH, list_cap, int(layer->max_nch));
    if (lds > 160 * 1024) return PBL_ERR_UNSUPPORTED;
    pbl_layer L = *layer;
    const _Float16* xp = static_cast<const _Float16*>(x);
    void* argv[] = {&L, &xp, &y, &M, &y_f32, &list_cap};
    const bool sf = layer->flags & PBL_FLAG_SAL_F16;
#define PBL_MFMA_PICK(W) (M <= 16 ? (sf ? reinterpret_cast<const void*>(pbl_mfma_kernel<1, true, W>) : reinterpret_cast<const void*>(pbl_mfma_kernel<1, false, W>)) \
                                   : (sf ? reinterpret_cast<const void*>(pbl_mfma_kernel<2, true, W>) : reinterpret_cast<const void*>(pbl_mfma_kernel<2, false, W>)))
    const void* k = wpg == 2 ? PBL_MFMA_PICK(2) : PBL_MFMA_PICK(4);
#undef PBL_MFMA_PICK
    if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess)
        return PBL_ERR_LAUNCH;
    return hipLaunchKernel(k, dim3(layer->NRB), dim3(wpg * GW), argv, lds, static_cast<hipStream_t>(stream)) == hipSuccess
               ? PBL_OK : PBL_ERR_LAUNCH;
}
