// pbl_gemm.hip -- small-batch GEMM (1 <= M <= 32 tokens) straight from the PBL1 packed format, on the
// matrix cores.  The GEMV streams the weights once per 4 tokens; this kernel streams them ONCE for up to
// 32 tokens (batched decode, short prefill: BASELINE config 4).  Replaces F.linear(x, W_fq, b) of the
// reference (quant/outlier_quantizer.py:105, gptq_pb/eval_ppl_utils.py:59-60).  Layers with or without column groups.
//
// Work decomposition (round 2; round 1 counting-sorted every record's chunks per launch and loaded x per wave):
//   * a workgroup = 4 waves = 4 consecutive 16-row records; all four walk the SAME columns, slab by slab
//     (256 columns), so the slab of x (up to 32 tokens x 256 columns, fp16) is staged in LDS ONCE per workgroup
//     (double buffered, one __syncthreads per slab) and every B fragment is a conflict-free ds_read_b128;
//   * optional K split: blockIdx.y takes a contiguous range of slabs and writes an fp32 partial y; a second small
//     kernel adds the partials in a fixed order (deterministic).  Layers with few records (N = 4096: 256) would
//     otherwise leave three quarters of the SIMDs idle.
// Per slab a wave
//   * drops the slab's two sign-plane dwords per lane into LDS (as is and << 8 for rows 8..15) and reads them back
//     as b128 (4 dwords = 8 columns x 16 rows, broadcast over the 16 row-lanes); ONE v_and_or per dword turns them
//     into a CLASS-CODED fp16 A fragment -- the GEMV's (w & M_c) | C_c trick indexed by the lane's own row;
//   * finds its record's salient chunks of the slab through the packer's slab index (include/pbl.h: no sort, no
//     search): the 4 lanes of a row (row = lane / 4) take the row's chunks one after the other, each lane a quarter
//     (4 entries: short dependent chains, no idle lanes at low density), and write them into the fp16 tile St[16][256] -- the code as 1024 + q (0x6400 | q: one OR, exact), or for
//     fp16 checkpoints the double-rounded fp16 weight; entries left or right of the slab are clamped into pad columns;
//   * derives the salient MASK fragment from the tile it has just read: v_pk_min_u16(s, 0x3C00) for codes (every
//     stored half is >= 0x6400), min(s, 1) * 0x3C00 for fp16 values (a salient of value 0 is stored as -0, so every
//     stored half has a bit set); no byte mask tile, no second scatter;
//   * runs v_mfma_f32_16x16x32_f16:  accW += W.x, accS += St.x, accM += Mask.x;
//   * clears St with nine ds_write_b128.
// X, the plain sum of a token's x over the split's columns, is the same for every record: the threads that stage x
// add up what they stage (v_dot2 with ones) and share the 32 sums through LDS at the end -- no fourth MFMA.
// Decode in fp32 (linear in the accumulators, so K-split partials simply add):
//   D = A_c*accW - B_c*X,  S = accM,  Q = accS - 1024*S (codes),
//   y = alpha*D + mu*X + [ss*(Q - sz*S) | accS] - hi*S  (+ exceptions + bias in split 0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

#define GW 64
#define SLAB PBL_SLAB_COLS
#define SSTR (SLAB + 8)      // halves per tile row: 528 B, so the 16 row-lanes of a b128 fragment read hit distinct banks
#define WPG 4                // waves (= records) per workgroup
#define PBL_NO_CHUNK (1 << 20)
// performance-analysis hook (tools/ablate_mfma.sh): bit 1 no scatter, 2 no MFMA, 3 no clears, 4 no x staging.
// 0 in every shipped build.
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

// One salient chunk in registers: its 16 byte steps, 16 codes, first column.  Two lane mappings (template parameter Q4):
//   Q4 = false  lane (row, slot) owns whole chunks slot, slot + 4, ... of its row: 16 entries per lane and pass.  Best when a
//               row has several chunks per slab (config 4 at 20 % salients: ~4): every lane is busy, 8 chunks per row are
//               prefetched a slab ahead.
//   Q4 = true   the 4 lanes of a row share ONE chunk, a quarter (4 entries) each: short dependent chains and no idle lanes
//               when a row has only 1-2 chunks per slab (<= ~12 % salients); measured -5 % at 4096^2 / 10 %, but +16 % on
//               config 4, where the round count is the MAXIMUM chunk count over the 16 rows.
struct ChunkRegs {
    u32x4 d4, q4;
    int col0;               // PBL_NO_CHUNK: nothing to do
};

struct MfmaArgs {
    pbl_layer L;
    const _Float16* x;      // [M, K]
    void* y;                // [M, N] fp16 / fp32 (KS == 1)
    float* part;            // [KS][M][N] fp32 (KS > 1)
    int M, y_f32, KS, sps;  // sps: slabs per K split
    int gshift;             // log2(columns per group) (GRP kernels)
};

constexpr size_t MFMA_WAVE_BYTES = 16 + size_t(16) * SSTR * 2 + 2 * 512 + 256;   // dump slot + St + Wp + Wp << 8 + row params
// One x tile instead of two for <= 16 tokens: 48 KB of LDS instead of 56, i.e. THREE workgroups per CU instead of two, for
// one more barrier per slab; the K split then aims at 768 workgroup slots.  Measured (gpurun_out/s24, 8-16 tokens):
// 11008x4096 29.2 -> 23.0 us, 13824x5120 40.6 -> 36.8, 4096x11008 29.0 -> 27.1, 4096^2 unchanged.
#ifndef PBL_MFMA_X1
#define PBL_MFMA_X1 1
#endif
__host__ __device__ constexpr int mfma_xbufs(int ntb) { return (PBL_MFMA_X1 && ntb == 1) ? 1 : 2; }
__host__ __device__ constexpr size_t mfma_lds_bytes(int ntb) {
    return size_t(mfma_xbufs(ntb)) * 16 * ntb * SSTR * 2 + WPG * MFMA_WAVE_BYTES + 128 + 512;   // + Xsum[32] + Xh[2][2][32]
}

// GRP: the layer has column groups (G > 1, a power-of-two number of columns >= 128 each): the binarized part and the salient
// mask are folded into per-row totals at every group boundary with that group's (hi, lo), exactly as the column-group GEMV does
// (pbl_kernels.hip); X is then needed per 128-column half slab and goes through LDS (Xh).
template <int NTB, bool SF, bool Q4, bool GRP>
__global__ __launch_bounds__(WPG * GW) void pbl_mfma_kernel(MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    constexpr int XT = 16 * NTB;                   // token rows of the x tile
    constexpr int XB = mfma_xbufs(NTB);            // x tiles in LDS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const pbl_layer& L = a.L;
    const int K = int(L.K), P = int(L.P), M = a.M;
    const int NS = (K + SLAB - 1) / SLAB;
    const uint32_t rb_raw = blockIdx.x * WPG + wave;
    const bool rec_ok = rb_raw < L.NRB;
    const uint32_t rb = rec_ok ? rb_raw : L.NRB - 1;   // a surplus wave mirrors the last record: uniform control flow, no store
    const int ks = int(blockIdx.y);
    const int s0 = ks * a.sps, s1 = min(s0 + a.sps, NS);

    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    const uint8_t* rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    const int nfull = __builtin_amdgcn_readfirstlane(info.y), ntail = __builtin_amdgcn_readfirstlane(info.z);
    const int nexc = __builtin_amdgcn_readfirstlane(info.w), nch = nfull + ntail;
    const uint32_t nchu = uint32_t(nch);
    const uint32_t tiles_off = GRP ? PBL_TILES_OFF(L.G) : PBL_TILES_OFF(1u);
    const uint8_t* sal = rec + tiles_off + uint32_t(P) * 1024u;
    const pbl_rowparams* params = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
    const pbl_rowinfo* rinfo = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF);
    const u32x4* tiles = reinterpret_cast<const u32x4*>(rec + tiles_off) + lane;
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const u32x4* codep = reinterpret_cast<const u32x4*>(sal + PBL_SAL_CODE_OFF(nchu));
    constexpr bool has_crow = SF || GRP;           // per-chunk row ids exist for fp16 checkpoints and for column groups
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));
    const uint32_t* slabtab = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_SLAB_OFF(nchu, uint32_t(ntail), uint32_t(nexc), has_crow));

    // LDS: x tiles [2][XT][SSTR] fp16 (shared); per wave St[16][SSTR] fp16, Wp[2][64] / Wp8[2][64] dwords (the slab's sign
    // plane, [128-column sub-block][lane], as is and << 8), the record's row params; Xsum[32] (shared, at the end)
    _Float16* Xs = reinterpret_cast<_Float16*>(smem_g);
    char* wbase = smem_g + size_t(XB) * XT * SSTR * 2 + size_t(wave) * MFMA_WAVE_BYTES;
    _Float16* St = reinterpret_cast<_Float16*>(wbase + 16);   // 16 bytes in front: where row 0's out-of-slab entries are dumped
    uint32_t* Wp = reinterpret_cast<uint32_t*>(wbase + 16 + size_t(16) * SSTR * 2);
    uint32_t* Wp8 = Wp + 128;
    float4* prm = reinterpret_cast<float4*>(Wp8 + 128);                        // [16] {hi, lo, sscale, szero}
    float* Xsum = reinterpret_cast<float*>(smem_g + size_t(XB) * XT * SSTR * 2 + WPG * MFMA_WAVE_BYTES);
    float* Xh = Xsum + 32;                          // [buf][half slab][token] (GRP)

    const int row_a = lane & 15, kblk = lane >> 4;   // fragment coordinates: A row / B token, 8-column block
    const int rho_s = lane >> 2, slot = lane & 3;    // scatter coordinates: the lane's row and its chunk slot

    // ---- staging of x: thread -> (token, 16-byte column chunk) of the slab ------------------------------------
    u32x4 xr[2 * NTB];
    auto load_x = [&](int s) {
#pragma unroll
        for (int j = 0; j < 2 * NTB; ++j) {
            const int idx = tid + j * (WPG * GW), tok = idx >> 5, col = s * SLAB + (idx & 31) * 8;
            u32x4 v = {0, 0, 0, 0};
            if (!(PBL_MFMA_ABLATE & 16) && tok < M && col < K) v = *reinterpret_cast<const u32x4*>(a.x + size_t(tok) * K + col);
            xr[j] = v;
        }
    };
    float xsum[2 * NTB];                               // this thread's share of X[token (tid >> 5) + 8 j]
#pragma unroll
    for (int j = 0; j < 2 * NTB; ++j) xsum[j] = 0.f;
    auto store_x = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2 * NTB; ++j) {
            const int idx = tid + j * (WPG * GW), tok = idx >> 5;
            *reinterpret_cast<u32x4*>(Xs + (size_t(buf) * XT + tok) * SSTR + (idx & 31) * 8) = xr[j];
            float h = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (through a scalar: hipcc 7.2 miscompiles __builtin_bit_cast applied directly to a vector ELEMENT -- it
                // reads element 0 whatever the subscript)
                const uint32_t w = xr[j][q];
                h = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2v, w), h2v{_Float16(1.f), _Float16(1.f)}, h, false);
            }
            if constexpr (GRP) {   // 16 consecutive lanes staged one token's 128-column half slab
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) h += __shfl_xor(h, d, GW);
                if ((tid & 15) == 0) Xh[(buf * 2 + ((tid >> 4) & 1)) * 32 + tok] = h;
            } else {
                xsum[j] += h;
            }
        }
    };

    // ---- the lane's row: chunk ranges per slab from the packer's slab index ---------------------------------
    // Memory latency is what bounds this kernel (two workgroups per CU), so nothing the inner loop needs is loaded
    // where it is used: slab-index entries are fetched three slabs ahead, chunk data (two passes = 8 chunks per row)
    // and the x slab one slab ahead, all issued at the top of an iteration with no dependent wait.
    const pbl_rowinfo ri = rinfo[rho_s];
    const uint32_t* tabrow = slabtab + rho_s * NS;
    auto tab = [&](int s) -> uint32_t { return (s >= 0 && s < NS) ? tabrow[s] : 0u; };
    struct Seq { int fb, fn, tb, tn; };
    auto seq_of = [&](uint32_t pe, uint32_t e) -> Seq {          // chunks of the row that overlap a slab: entries of slab s-1, s
        Seq q;
        q.fb = int(PBL_SLAB_FE(pe)) - int(PBL_SLAB_FBACK(e)); q.fn = int(PBL_SLAB_FE(e)) - q.fb;
        q.tb = int(PBL_SLAB_TE(pe)) - int(PBL_SLAB_TBACK(e)); q.tn = int(PBL_SLAB_TE(e)) - q.tb;
        return q;
    };
    // j-th chunk of the row's sequence for the slab (its full chunks, then its tail chunks): the same chunk for the row's 4 lanes
    // round `rnd` of the lane's work on the slab: chunk rnd of the row (Q4: shared by the row's 4 lanes) or chunk slot + 4 rnd
    auto load_chunk = [&](int rnd, const Seq& sq) -> ChunkRegs {
        ChunkRegs r;
        r.col0 = PBL_NO_CHUNK; r.d4 = u32x4{0, 0, 0, 0}; r.q4 = u32x4{0, 0, 0, 0};
        const int j = Q4 ? rnd : slot + 4 * rnd;
        int c = -1;
        if (j < sq.fn) c = int(ri.start) + sq.fb + j;
        else if (j - sq.fn < sq.tn) c = nfull + int(ri.tailidx) + sq.tb + (j - sq.fn);
        if (c >= 0) {
            if (PBL_MFMA_ABLATE & 64) { r.d4 = u32x4{0x08060402u, 0x0a0c0e04u, 0x02040608u, 0x10020406u}; r.q4 = u32x4{uint32_t(c), 77u, 99u, 3u}; r.col0 = (c * 37) & 0xFFF; }   // no memory traffic
            else { r.d4 = deltap[c]; r.q4 = codep[c]; r.col0 = int(col0p[c]); }
        }
        return r;
    };
    // rounds the slowest row of the slab needs
    auto rounds_left = [&](int rnd, const Seq& sq) -> bool { return __any((Q4 ? rnd : slot + 4 * rnd) < sq.fn + sq.tn); };
    const float4 prs = reinterpret_cast<const float4*>(params)[rho_s];        // the scatter lane's row params (SF)
    // One chunk (Q4: the lane's quarter of it) -> St of the slab starting at column cb.  Tail padding repeats the last entry
    // (PBL_FLAG_TAIL_REPEAT), so a writer needs no count.  Deltas are stored doubled = byte steps in an fp16 row, so one
    // SDWA add per entry advances the running LDS byte address; ONE v_med3_i32 sends entries outside the slab to a dump
    // slot: left of the slab into the 8 pad columns of the previous tile row (row 0: the 16 bytes in front of the tile),
    // right of it into the row's own pad columns -- eight different ones per side, chosen by the lane, so that clamped
    // writes do not pile up on one address.
    // SF: the tile holds MINUS the weight, u = fl16(fl32((-ss) * (q - sz))): the rounding is symmetric, so u = -w exactly,
    // and a weight of value zero ((q - sz) = +0 times a negative factor) comes out as -0 = 0x8000 -- every stored half has a
    // bit set and the mask is "any bit set", without a compare per entry.  The epilogue negates the accumulated sum.  (The
    // packers never emit an entry whose u would be +0: zero scale, or a product that underflows from the positive side.)
    const int st_row = int(uint32_t(reinterpret_cast<uintptr_t>(St))) + rho_s * (SSTR * 2);   // LDS byte address of the lane's tile row
    const int clamp_lo = st_row - 2 * (1 + (lane & 7)), clamp_hi = st_row + 2 * (SLAB + (lane & 7));
    float nss = -prs.z;
    // (opaque to the optimiser: it must stay a multiplication by a NEGATIVE factor -- rewritten as ss * (sz - q) the product of a
    // zero-valued salient would be +0, an empty tile position)
    asm volatile("" : "+v"(nss));
    uint32_t abl_acc = 0;   // ablation builds only
    auto store_half = [&](int addr, uint32_t v, bool hi) {
        if (PBL_MFMA_ABLATE & 32) { abl_acc ^= uint32_t(addr) ^ v; return; }     // everything but the LDS write
        if (hi) asm volatile("ds_write_b16_d16_hi %0, %1" :: "v"(addr), "v"(v) : "memory");
        else asm volatile("ds_write_b16 %0, %1" :: "v"(addr), "v"(v) : "memory");
    };
    auto scatter = [&](const ChunkRegs& r, int cb) {
        if ((PBL_MFMA_ABLATE & 2) || r.col0 == PBL_NO_CHUNK) return;
        int run = st_row + 2 * (r.col0 - cb);
        // entries e0, e0 + 1 of the lane's part: byte steps d0, d1, codes q0, q1
        auto put2 = [&](uint32_t d0, uint32_t d1, uint32_t q0, uint32_t q1) {
            int a0, a1;
            run += int(d0);
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(a0) : "v"(run), "v"(clamp_lo), "v"(clamp_hi));
            run += int(d1);
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(a1) : "v"(run), "v"(clamp_lo), "v"(clamp_hi));
            uint32_t pair;
            if constexpr (SF) {
                h2v w;
                w.x = round_f16_twice(nss * (float(q0) - prs.w));
                w.y = round_f16_twice(nss * (float(q1) - prs.w));
                pair = __builtin_bit_cast(uint32_t, w);
            } else {
                pair = 0x64006400u | q0 | (q1 << 16);            // fp16 1024 + q, exact
            }
            store_half(a0, pair, false);
            store_half(a1, pair, true);
        };
        auto byte_of = [](uint32_t w, int k) -> uint32_t { return (w >> (8 * k)) & 0xFFu; };
        if constexpr (Q4) {
            uint32_t pre = 0;
            pre = slot > 0 ? __builtin_amdgcn_sad_u8(r.d4[0], 0u, pre) : pre;
            pre = slot > 1 ? __builtin_amdgcn_sad_u8(r.d4[1], 0u, pre) : pre;
            pre = slot > 2 ? __builtin_amdgcn_sad_u8(r.d4[2], 0u, pre) : pre;
            const uint32_t dd = slot == 0 ? r.d4[0] : (slot == 1 ? r.d4[1] : (slot == 2 ? r.d4[2] : r.d4[3]));
            const uint32_t qq = slot == 0 ? r.q4[0] : (slot == 1 ? r.q4[1] : (slot == 2 ? r.q4[2] : r.q4[3]));
            run += int(pre);
            put2(byte_of(dd, 0), byte_of(dd, 1), byte_of(qq, 0), byte_of(qq, 1));
            put2(byte_of(dd, 2), byte_of(dd, 3), byte_of(qq, 2), byte_of(qq, 3));
        } else {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                // (through scalars: hipcc 7.2 miscompiles some element-wise uses of a vector ELEMENT in place)
                const uint32_t dw = r.d4[e >> 2], qw = r.q4[e >> 2];
                put2(byte_of(dw, e & 3), byte_of(dw, (e & 3) + 1), byte_of(qw, e & 3), byte_of(qw, (e & 3) + 1));
            }
        }
    };

    v4f accW[NTB], accS[NTB], accM[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) { accW[t] = accS[t] = accM[t] = v4f{0.f, 0.f, 0.f, 0.f}; }
    // GRP: per-row totals over the finished groups, the salient mask sum over all groups, X of the open group; class constants
    // of the lane's four OUTPUT rows; (hi, lo) of the groups the current slab's two halves lie in, fetched a slab ahead
    v4f tot[NTB], Stot[NTB];
    float Xg[NTB], Ar[4], Br[4];
    float2 hlc[2][4], hln[2][4];
#pragma unroll
    for (int t = 0; t < NTB; ++t) { tot[t] = Stot[t] = v4f{0.f, 0.f, 0.f, 0.f}; Xg[t] = 0.f; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t cu;
        class_consts_g((4 * (lane >> 4) + r) & 7, Ar[r], Br[r], cu);
        hlc[0][r] = hlc[1][r] = hln[0][r] = hln[1][r] = make_float2(0.f, 0.f);
    }
    const float2* ghl = reinterpret_cast<const float2*>(rec + PBL_REC_GHL_OFF);
    const int Gm1 = int(L.G) - 1;
    auto load_hl = [&](int s, float2 (&dst)[2][4]) {
        if constexpr (GRP) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int g = min((s * SLAB + hf * 128) >> a.gshift, Gm1);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[hf][r] = ghl[(4 * (lane >> 4) + r) * int(L.G) + g];
            }
        }
    };
    auto fold = [&](const float2 (&hl)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float alpha = 0.5f * (hl[r].x - hl[r].y), mu = 0.5f * (hl[r].x + hl[r].y);
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const float D = fmaf(Ar[r], accW[t][r], -(Br[r] * Xg[t]));
                tot[t][r] += fmaf(alpha, D, fmaf(mu, Xg[t], -(hl[r].x * accM[t][r])));
                Stot[t][r] += accM[t][r];
            }
        }
#pragma unroll
        for (int t = 0; t < NTB; ++t) { accW[t] = accM[t] = v4f{0.f, 0.f, 0.f, 0.f}; Xg[t] = 0.f; }
    };
    // the lane's row decides its bit (rows >= 8 read the pre-shifted copy, then bit 8 + class in each half-word) and class
    float Acl, Bcl;
    uint32_t Ccl;
    class_consts_g(row_a & 7, Acl, Bcl, Ccl);
    const uint32_t Mcl = 0x01000100u << (row_a & 7);
    const uint32_t* Wsel = row_a >= 8 ? Wp8 : Wp;
    auto frag = [&](const u32x4 d) -> v8h {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (d[q] & Mcl) | Ccl;
        return __builtin_bit_cast(v8h, o);
    };
    uint32_t c_one2 = 0x3C003C00u, c_int1 = 0x00010001u;   // fp16x2 (1.0, 1.0) and u16x2 (1, 1) in VGPRs
    asm volatile("" : "+v"(c_one2), "+v"(c_int1));
    // salient mask {0, 1.0} from the tile fragment itself
    auto mfrag = [&](const u32x4 s) -> v8h {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (SF) {   // any bit set (a zero VALUE is stored as -0): min(half, 1) * 0x3C00, two packed u16 ops
                uint32_t t;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(s[q]), "v"(c_int1));
                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o[q]) : "v"(t), "v"(c_one2));
            } else {              // stored halves are 0 or >= 0x6400: unsigned min with fp16 1.0.  Inline asm: hipcc 7.2
                                  // folds __builtin_elementwise_min over the four dwords of a vector into the FIRST one
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(o[q]) : "v"(s[q]), "v"(c_one2));
            }
        }
        return __builtin_bit_cast(v8h, o);
    };

    // ---- prologue -------------------------------------------------------------------------------------------
    if (tid < 16 * WPG) {   // every wave's row params -> its LDS slot (thread t: wave t / 16, row t % 16)
        const uint32_t wr = min(blockIdx.x * WPG + uint32_t(tid >> 4), L.NRB - 1);
        const uint4 winfo = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[wr];
        const float4* wp = reinterpret_cast<const float4*>(blob + size_t(winfo.x) * 16 + PBL_REC_PARAMS_OFF);
        reinterpret_cast<float4*>(smem_g + size_t(XB) * XT * SSTR * 2 + size_t(tid >> 4) * MFMA_WAVE_BYTES + 16 + size_t(16) * SSTR * 2 + 1024)[tid & 15] = wp[tid & 15];
    }
    u32x4 t_cur = {0, 0, 0, 0}, t_next = {0, 0, 0, 0};
    ChunkRegs cA, cB, cC, nA, nB, nC;                  // rounds 0, 1, 2 of the current / next slab
    cA.col0 = cB.col0 = cC.col0 = nA.col0 = nB.col0 = nC.col0 = PBL_NO_CHUNK;
    cA.d4 = cB.d4 = cC.d4 = nA.d4 = nB.d4 = nC.d4 = u32x4{0, 0, 0, 0};
    cA.q4 = cB.q4 = cC.q4 = nA.q4 = nB.q4 = nC.q4 = u32x4{0, 0, 0, 0};
    uint32_t eP = 0, e0 = 0, e1 = 0, e2 = 0;           // slab-index entries of slabs s-1, s, s+1, s+2
    Seq sq = {0, 0, 0, 0};
    if (s0 < s1) {
        t_cur = __builtin_nontemporal_load(tiles + (s0 >> 1) * 64);
        load_x(s0);
        eP = tab(s0 - 1); e0 = tab(s0); e1 = tab(s0 + 1); e2 = tab(s0 + 2);
        sq = seq_of(eP, e0);
        cA = load_chunk(0, sq);
        cB = load_chunk(1, sq);
        if (Q4) cC = load_chunk(2, sq);
        load_hl(s0, hlc);
    }
    for (int i = lane; i < int(size_t(16) * SSTR * 2 / 16); i += GW) reinterpret_cast<u32x4*>(St)[i] = u32x4{0, 0, 0, 0};
    if (s0 < s1) store_x(0);

    for (int s = s0; s < s1; ++s) {
        const int buf = XB == 2 ? (s - s0) & 1 : 0, half = s & 1, cb = s * SLAB;
        const bool more = s + 1 < s1;
        // x of THIS slab, loaded during the previous iteration, goes to its LDS tile first: the only wait on memory in
        // the loop then is for loads that have had a whole iteration to land.  (Tile `buf` was last read two slabs ago
        // and every wave has passed the previous slab's barrier since.)
        if (s > s0) {
            if (XB == 1) __syncthreads();          // a single x tile: every wave must be done with the previous slab's fragments
            store_x(buf);
        }
        // everything slab s+1 needs from memory, issued now, consumed a slab later
        const Seq sqn = seq_of(e0, e1);
        uint32_t e3 = 0;
        if (more) {
            load_x(s + 1);
            nA = load_chunk(0, sqn);
            nB = load_chunk(1, sqn);
            if (Q4) nC = load_chunk(2, sqn);
            e3 = tab(s + 3);
            load_hl(s + 1, hln);
            if (half) t_next = __builtin_nontemporal_load(tiles + ((s + 1) >> 1) * 64);
        }
        {   // this slab's two sign-plane dwords -> Wp[sub-block][lane], as is and << 8
            const uint32_t w0 = half ? t_cur[2] : t_cur[0], w1 = half ? t_cur[3] : t_cur[1];
            Wp[lane] = w0; Wp[64 + lane] = w1; Wp8[lane] = w0 << 8; Wp8[64 + lane] = w1 << 8;
        }
        // salient entries of the slab -> St: the first rounds (Q4: 3 chunks per row, else 8) were loaded one slab ahead; rows
        // with more chunks in a slab fetch the rest on demand
        scatter(cA, cb);
        scatter(cB, cb);
        if (Q4) scatter(cC, cb);
        for (int rnd = Q4 ? 3 : 2; rounds_left(rnd, sq); ++rnd) scatter(load_chunk(rnd, sq), cb);
        __syncthreads();   // x tile `buf` complete (written one slab ago); everybody is done reading tile buf ^ 1; St / Wp ordered
        const _Float16* xt = Xs + size_t(buf) * XT * SSTR;
        // all 8 k-steps, branch free: right of K the x tile holds zeros (and the plane / St nothing), so a ragged last
        // slab just adds zeros.  The fragments of k-step k+1 are read from LDS before the MFMAs of k-step k are issued
        // (two waves per SIMD do not hide an LDS round trip per k-step by themselves).
        struct Frags { u32x4 sd, wd; v8h bx[NTB]; };
        auto read_frags = [&](int k8) -> Frags {
            Frags f;
            f.sd = *reinterpret_cast<const u32x4*>(St + row_a * SSTR + k8 * 32 + kblk * 8);
            f.wd = *reinterpret_cast<const u32x4*>(Wsel + (k8 >> 2) * 64 + (k8 & 3) * 16 + kblk * 4);
#pragma unroll
            for (int t = 0; t < NTB; ++t) f.bx[t] = *reinterpret_cast<const v8h*>(xt + (t * 16 + row_a) * SSTR + k8 * 32 + kblk * 8);
            return f;
        };
        Frags fcur = read_frags(0);
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            Frags fnext = fcur;
            if (k8 + 1 < 8) fnext = read_frags(k8 + 1);
            const v8h aW = frag(fcur.wd);
            const v8h aS = __builtin_bit_cast(v8h, fcur.sd);
            const v8h aM = mfrag(fcur.sd);
#pragma unroll
            for (int t = 0; t < ((PBL_MFMA_ABLATE & 4) ? 0 : NTB); ++t) {
                accW[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aW, fcur.bx[t], accW[t], 0, 0, 0);
                accS[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aS, fcur.bx[t], accS[t], 0, 0, 0);
                accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aM, fcur.bx[t], accM[t], 0, 0, 0);
            }
            fcur = fnext;
            if constexpr (GRP) {
                if ((k8 & 3) == 3) {   // a 128-column half slab is done: its X, and the fold if its group ends here
                    const int hf = k8 >> 2;
#pragma unroll
                    for (int t = 0; t < NTB; ++t) Xg[t] += Xh[(buf * 2 + hf) * 32 + t * 16 + row_a];
                    const int cend = cb + 128 * (hf + 1);
                    if (!(cend & ((1 << a.gshift) - 1)) || (hf == 1 && !more)) fold(hlc[hf]);
                }
            }
        }
        if constexpr (GRP) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 4; ++r) hlc[hf][r] = hln[hf][r];
        }
        if (!(PBL_MFMA_ABLATE & 8)) {
            for (int i = lane; i < int(size_t(16) * SSTR * 2 / 16); i += GW) reinterpret_cast<u32x4*>(St)[i] = u32x4{0, 0, 0, 0};
        }
        cA = nA; cB = nB; cC = nC; sq = sqn;
        eP = e0; e0 = e1; e1 = e2; e2 = e3;
        if (half) t_cur = t_next;
        // the tile is written as halves / dwords and read as 16-byte vectors: keep the compiler from moving the next
        // slab's stores across this slab's (type-based alias analysis would allow it; the LDS itself is in order)
        asm volatile("" ::: "memory");
    }

    // ---- X (the plain sum of x over this split's columns): 32 consecutive lanes staged one token's columns ----
#pragma unroll
    for (int j = 0; j < 2 * NTB; ++j) {
        float v = xsum[j];
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, GW);
        if ((tid & 31) == 0) Xsum[(tid >> 5) + 8 * j] = v;
    }
    __syncthreads();
    if (!rec_ok) return;
    if ((PBL_MFMA_ABLATE & 32) && abl_acc == 0x12345678u) a.part[0] = 1.f;

#pragma unroll
    for (int r = 0; r < 4; ++r) {                    // lane holds token (lane & 15) of each token block, row 4*(lane >> 4) + r
        const int rho = 4 * kblk + r;
        const uint32_t row = rb * 16 + rho;
        if (row >= L.N) continue;
        const float4 pr = prm[rho];                 // {hi, lo, sscale, szero}
        float A, B;
        uint32_t Cunused;
        class_consts_g(rho & 7, A, B, Cunused);
        const float alpha = 0.5f * (pr.x - pr.y), mu = 0.5f * (pr.x + pr.y);
        const float bias = (ks == 0 && L.bias) ? L.bias[row] : 0.f;
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            const int tok = t * 16 + row_a;
            if (tok >= M) continue;
            float e = 0.f;
            if (ks == 0) {
                for (int k = 0; k < nexc; ++k) {
                    const uint2 ex = exc[k];
                    if (int(ex.x >> 16) == rho) {
                        const uint32_t col = ex.x & 0xFFFFu;
                        const float hv = GRP ? ghl[rho * int(L.G) + int(col >> a.gshift)].x : pr.x;
                        e += (__builtin_bit_cast(float, ex.y) - hv) * float(a.x[size_t(tok) * K + col]);
                    }
                }
            }
            float out;
            if constexpr (GRP) {     // the groups are folded already; what is left is the salient weights themselves
                const float Sv = accS[t][r], S = Stot[t][r];
                const float salv = SF ? -Sv : pr.z * fmaf(-pr.w, S, fmaf(-1024.f, S, Sv));   // (SF: the tile holds minus the weight)
                out = tot[t][r] + salv + e + bias;
            } else {
                const float X = Xsum[t * 16 + row_a];
                const float Wv = accW[t][r], Sv = accS[t][r], S = accM[t][r];
                const float D = fmaf(A, Wv, -(B * X));
                float salv;
                if constexpr (SF) salv = fmaf(-pr.x, S, -Sv);      // the tile holds minus the weight
                else salv = fmaf(pr.z, fmaf(-pr.w, S, fmaf(-1024.f, S, Sv)), -(pr.x * S));
                out = fmaf(alpha, D, fmaf(mu, X, salv)) + e + bias;
            }
            if (a.KS > 1) a.part[(size_t(ks) * M + tok) * L.N + row] = out;
            else if (a.y_f32) static_cast<float*>(a.y)[size_t(tok) * L.N + row] = out;
            else static_cast<_Float16*>(a.y)[size_t(tok) * L.N + row] = _Float16(out);
        }
    }
}

// y = sum over the K splits, in split order (deterministic); 4 elements per thread (MN is a multiple of 4 whenever N is,
// the scalar tail covers the rest)
__global__ __launch_bounds__(256) void pbl_mfma_reduce(const float* __restrict__ part, void* __restrict__ y, int KS, size_t MN, int y_f32) {
    const size_t i = (size_t(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    if (i + 4 <= MN && !(MN & 3)) {
        // the partial outputs of up to 8 splits are requested together (one memory latency, not one per split) and added in
        // split order
        float4 s = *reinterpret_cast<const float4*>(part + i);
        for (int k0 = 1; k0 < KS; k0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = k0 + j < KS ? *reinterpret_cast<const float4*>(part + size_t(k0 + j) * MN + i) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + j < KS) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
            }
        }
        if (y_f32) *reinterpret_cast<float4*>(static_cast<float*>(y) + i) = s;
        else {
            _Float16* o = static_cast<_Float16*>(y) + i;
            o[0] = _Float16(s.x); o[1] = _Float16(s.y); o[2] = _Float16(s.z); o[3] = _Float16(s.w);
        }
        return;
    }
    for (size_t j = i; j < MN && j < i + 4; ++j) {
        float s = part[j];
        for (int k = 1; k < KS; ++k) s += part[size_t(k) * MN + j];
        if (y_f32) static_cast<float*>(y)[j] = s;
        else static_cast<_Float16*>(y)[j] = _Float16(s);
    }
}

// K splits so that ONE round of workgroups fills the chip: 256 CUs x 2 resident workgroups (LDS bound).  A second,
// partly filled round costs a whole workgroup time, and every workgroup pays a prologue of ~3 dependent memory
// latencies, so fewer, longer workgroups win; at least 2 slabs per split.
void pick_split(const pbl_layer* L, int& KS, int& sps, int ntb = 2) {
    // three resident workgroups per CU need <= 170 VGPRs per wave as well: the column-group kernels (182-234) stay at two
    const bool three = mfma_xbufs(ntb) == 1 && L->G == 1;
    const int NS = int((L->K + SLAB - 1) / SLAB);
    const int groups = int((L->NRB + WPG - 1) / WPG);
#ifndef PBL_MFMA_SLOTS
#define PBL_MFMA_SLOTS 512
#endif
#ifndef PBL_MFMA_MIN_SLABS
#define PBL_MFMA_MIN_SLABS 2
#endif
    int ks = (three ? (PBL_MFMA_SLOTS * 3) / 2 : PBL_MFMA_SLOTS) / groups;
    if (ks > NS / PBL_MFMA_MIN_SLABS) ks = NS / PBL_MFMA_MIN_SLABS;
    if (ks < 1) ks = 1;
    sps = (NS + ks - 1) / ks;
    KS = (NS + sps - 1) / sps;
}

bool mfma_supported(const pbl_layer* layer, const void* x) {
    if (layer->G > 1) {   // column groups: a power-of-two number of columns, at least a half slab
        const uint32_t gs = layer->K / layer->G;
        if (gs * layer->G != layer->K || (gs & (gs - 1)) || gs < 128) return false;
    }
    return !(layer->K & 7) && !(reinterpret_cast<uintptr_t>(x) & 15) &&
           (layer->flags & PBL_FLAG_TAIL_REPEAT) && (layer->flags & PBL_FLAG_SLABS);
}

}  // namespace

extern "C" size_t pbl_mfma_workspace_bytes(const pbl_layer* layer, int M) {
    if (!layer || M < 1) return 0;
    int KS, sps;
    const int mb = M < 32 ? M : 32;
    pick_split(layer, KS, sps, mb <= 16 ? 1 : 2);
    return KS > 1 ? size_t(KS) * mb * layer->N * sizeof(float) : 0;
}

extern "C" int pbl_gemm_mfma_f16_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!layer || !layer->blob || !x || !y || M < 1 || M > 32) return PBL_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(layer->blob) & 15) return PBL_ERR_MISALIGNED;
    if (!mfma_supported(layer, x)) return PBL_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MfmaArgs a;
    a.L = *layer; a.x = static_cast<const _Float16*>(x); a.y = y; a.M = M; a.y_f32 = y_f32;
    pick_split(layer, a.KS, a.sps, M <= 16 ? 1 : 2);
    const size_t need = size_t(a.KS) * M * layer->N * sizeof(float);
    if (a.KS > 1 && (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))) {
        a.KS = 1; a.sps = int((layer->K + SLAB - 1) / SLAB);              // no workspace: one split
    }
    a.part = a.KS > 1 ? static_cast<float*>(workspace) : nullptr;
    const bool sf = layer->flags & PBL_FLAG_SAL_F16;
    const int ntb = M <= 16 ? 1 : 2;
    // lane mapping of the salient scatter (see ChunkRegs): four lanes per chunk when a row has at most ~2 chunks per slab
    const uint32_t NSl = (layer->K + SLAB - 1) / SLAB;
    const bool q4 = layer->max_nch <= 35u * NSl;                          // 16 rows x 2.2 chunks per row and slab
    const bool grp = layer->G > 1;
    a.gshift = grp ? 31 - __builtin_clz(layer->K / layer->G) : 31;
#define PBL_PICK2(NTB_, SF_, Q4_) (grp ? reinterpret_cast<const void*>(pbl_mfma_kernel<NTB_, SF_, Q4_, true>) : reinterpret_cast<const void*>(pbl_mfma_kernel<NTB_, SF_, Q4_, false>))
#define PBL_PICK(NTB_, SF_) (q4 ? PBL_PICK2(NTB_, SF_, true) : PBL_PICK2(NTB_, SF_, false))
    const void* k = ntb == 1 ? (sf ? PBL_PICK(1, true) : PBL_PICK(1, false)) : (sf ? PBL_PICK(2, true) : PBL_PICK(2, false));
#undef PBL_PICK
#undef PBL_PICK2
    const size_t lds = mfma_lds_bytes(ntb);
    if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess)
        return PBL_ERR_LAUNCH;
    void* argv[] = {&a};
    if (hipLaunchKernel(k, dim3((layer->NRB + WPG - 1) / WPG, a.KS), dim3(WPG * GW), argv, lds, st) != hipSuccess) return PBL_ERR_LAUNCH;
    if (a.KS > 1) {
        const float* part = a.part;
        size_t MN = size_t(M) * layer->N;
        int KS = a.KS;
        void* rv[] = {&part, &y, &KS, &MN, &y_f32};
        if (hipLaunchKernel(reinterpret_cast<const void*>(pbl_mfma_reduce), dim3(uint32_t((MN + 1023) / 1024)), dim3(256), rv, 0, st) != hipSuccess)
            return PBL_ERR_LAUNCH;
    }
    return PBL_OK;
}

extern "C" int pbl_gemm_mfma_f16(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* stream) {
    return pbl_gemm_mfma_f16_ws(layer, x, y, M, y_f32, nullptr, 0, stream);
}
