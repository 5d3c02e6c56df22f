// pbl_gemm_big.hip -- GEMM regime (more than 32 tokens: prefill, large batches) straight from the PBL1 packed format.
// Replaces F.linear(x, W_fq, b) over the dense fp16 fake-quant weight (gptq_pb/eval_ppl_utils.py:55-64: the reference's
// perplexity loop calls every nn.Linear with seq 2048 rows) WITHOUT the dense weight ever existing in HBM: round 1
// unpacked the layer into a transient workspace and ran a library GEMM on it.
//
// For layers packed from an fp16 checkpoint (PBL_FLAG_SAL_F16, G == 1): every weight is an fp16 number -- one of the
// row's two levels, the double-rounded fp16 salient value, or an explicit exception -- so the kernel rebuilds the EXACT
// fp16 weight tile in LDS and feeds it to v_mfma_f32_16x16x32_f16.  The arithmetic is that of an fp16 GEMM with fp32
// accumulation on the reference's own dense weight.
//
// Workgroup = 8 waves = 8 consecutive records (128 output rows) x 256 tokens; K is walked in half slabs of 128 columns.
//   A operand  As[2][128 rows][128 + 8] fp16 (double buffered): wave w EXPANDS its own record's 16 rows -- sign plane:
//              per (row, dword) shift / and / mad / v_perm_b32 picks {hi, lo} for two columns and one ds_write_b32
//              stores them; then the record's salient chunks of the slab (packer's slab index, as in pbl_gemm.hip) and
//              its exceptions are written over the tile -- while the OTHER buffer is being multiplied.
//   B operand  Xs[2][256 tokens][64] fp16, XOR-swizzled 16-byte units (conflict-free b128 fragment reads without
//              padding), staged through registers one 64-column sub-step ahead.
//   MFMA       waves as 2 (rows) x 4 (tokens): a wave owns 64 rows x 64 tokens = 16 accumulator tiles; per 32-column
//              k-step 4 A + 4 B fragment reads feed 16 MFMAs.
//   Overlap    expansion is VALU / LDS-write work, the product is matrix-core work.  The two waves that share a SIMD run
//              them in opposite order inside an iteration (waves 0-3 expand the next half slab BEFORE their MFMAs of a
//              sub-step, waves 4-7 AFTER), so one wave's VALU phase sits under the other's MFMA phase.  Both orders are
//              race free: the tile being expanded is not read until the iteration's last barrier.
//   Epilogue   accumulators -> LDS [token][row] -> contiguous 16-byte stores (a token's 128 rows are 256 contiguous bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GW 64
#define NWAVE 8
#define GB_ROWS 128
#define GB_TOK 256
#define GB_HS 128                 // columns per As buffer (half a slab)
#define GB_ASTR (GB_HS + 8)       // halves per As row: 272 B, the 16 row-lanes of a b128 read hit distinct banks
#define GB_XC 64                  // columns per x sub-step
#define GB_NO_CHUNK (1 << 20)
#define GB_AS_BYTES (size_t(GB_ROWS) * GB_ASTR * 2)
#define GB_XS_BYTES (size_t(GB_TOK) * GB_XC * 2)
#define GB_LDS (2 * GB_AS_BYTES + 2 * GB_XS_BYTES)

// performance-analysis hook (tools/build_variant.sh): bit 0 no expansion in the loop, 1 no x staging in the loop, 2 no MFMA,
// 3 no barriers in the loop.  0 in every shipped build (results are wrong otherwise).
#ifndef PBL_GEMM_ABLATE
#define PBL_GEMM_ABLATE 0
#endif

namespace {

__device__ __forceinline__ _Float16 round_f16_twice(float prod) {
    asm volatile("" : "+v"(prod));   // keep the fp32 product: the checkpoint value is double rounded
    return _Float16(prod);
}

struct ChunkRegs {
    u32x4 d4, q4;
    int col0;
};

struct GemmArgs {
    pbl_layer L;
    const _Float16* x;      // [M, K]
    _Float16* y;            // [M, N]
    int M;
};

__global__ __launch_bounds__(NWAVE * GW) void pbl_gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const pbl_layer& L = a.L;
    const int K = int(L.K), P = int(L.P), M = a.M;
    const int NS = (K + PBL_SLAB_COLS - 1) / PBL_SLAB_COLS;
    // XCD-aware work order (speed only): workgroup b runs on XCD b % 8; give every XCD a CONTIGUOUS range of the
    // token-tile-major work list, so the 32 workgroups resident on an XCD share one 256-token slab of x (2 MB at K = 4096:
    // it stays in that XCD's 4 MiB L2 instead of being fetched by all eight).  Bijective for any grid size.
    const uint32_t nrbk = (L.NRB + NWAVE - 1) / NWAVE, nwg = gridDim.x;
    const uint32_t xq = nwg >> 3, xr_ = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
#ifndef PBL_GEMM_XCD_MAP
#define PBL_GEMM_XCD_MAP 1
#endif
    const uint32_t wg = PBL_GEMM_XCD_MAP ? (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + xi : blockIdx.x;
    const uint32_t rowblk = wg % nrbk;
    const uint32_t rb_raw = rowblk * NWAVE + wave;
    const uint32_t rb = rb_raw < L.NRB ? rb_raw : L.NRB - 1;        // a surplus wave mirrors the last record; its rows are never stored
    const int tok0 = int(wg / nrbk) * GB_TOK;

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
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const u32x4* codep = reinterpret_cast<const u32x4*>(sal + PBL_SAL_CODE_OFF(nchu));
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), true));
    const uint32_t* slabtab = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_SLAB_OFF(nchu, uint32_t(ntail), uint32_t(nexc), true));

    _Float16* As = reinterpret_cast<_Float16*>(smem_b);                              // [2][128][GB_ASTR]
    _Float16* Xs = reinterpret_cast<_Float16*>(smem_b + 2 * GB_AS_BYTES);            // [2][256][64], 16-byte units XOR (token >> 1) & 7

    const int row_a = lane & 15, kblk = lane >> 4;      // fragment coordinates
    const int rho_s = lane >> 2, slot = lane & 3;       // scatter coordinates
    const int wr = wave >> 2, wc = wave & 3;            // this wave's 64-row x 64-token block of the workgroup tile
    const bool early = wave < 4;                        // expansion before (waves 0-3) or after (4-7) the sub-step's MFMAs

    // ---- the record's levels as fp16 pairs {hi : lo} per row, broadcast with v_readlane in the unrolled expansion ----
    uint32_t hilo_lane;
    {
        const pbl_rowparams pr = params[row_a];
        const uint32_t h = __builtin_bit_cast(uint16_t, _Float16(pr.hi)), l = __builtin_bit_cast(uint16_t, _Float16(pr.lo));
        hilo_lane = (h << 16) | l;
    }
    const pbl_rowinfo ri = rinfo[rho_s];
    const float4 prs = reinterpret_cast<const float4*>(params)[rho_s];
    const uint32_t* tabrow = slabtab + rho_s * NS;
    auto tab = [&](int s) -> uint32_t { return (s >= 0 && s < NS) ? tabrow[s] : 0u; };
    struct Seq { int fb, fn, tb, tn; };
    auto seq_of = [&](uint32_t pe, uint32_t e) -> Seq {
        Seq q;
        q.fb = int(PBL_SLAB_FE(pe)) - int(PBL_SLAB_FBACK(e)); q.fn = int(PBL_SLAB_FE(e)) - q.fb;
        q.tb = int(PBL_SLAB_TE(pe)) - int(PBL_SLAB_TBACK(e)); q.tn = int(PBL_SLAB_TE(e)) - q.tb;
        return q;
    };
    auto load_chunk = [&](int q, const Seq& sq) -> ChunkRegs {
        ChunkRegs r;
        r.col0 = GB_NO_CHUNK; r.d4 = u32x4{0, 0, 0, 0}; r.q4 = u32x4{0, 0, 0, 0};
        int c = -1;
        if (q < sq.fn) c = int(ri.start) + sq.fb + q;
        else if (q - sq.fn < sq.tn) c = nfull + int(ri.tailidx) + sq.tb + (q - sq.fn);
        if (c >= 0) { r.d4 = deltap[c]; r.q4 = codep[c]; r.col0 = int(col0p[c]); }
        return r;
    };
    const uint32_t padcol = uint32_t(GB_HS + (lane & 7));
    // all 16 entries of one chunk -> the tile rows of this wave's record, half slab starting at column cb
    auto scatter = [&](const ChunkRegs& r, int cb, uint16_t* arow) {
        if (r.col0 == GB_NO_CHUNK) return;
        int col = r.col0 - cb;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            col += int(((r.d4[e >> 2] >> (8 * (e & 3))) & 0xFFu) >> 1);
            const uint32_t cc = min(uint32_t(col), padcol);
            const uint32_t q = (r.q4[e >> 2] >> (8 * (e & 3))) & 0xFFu;
            arow[cc] = __builtin_bit_cast(uint16_t, round_f16_twice(prs.z * (float(q) - prs.w)));
        }
    };
    // sign plane + salients + exceptions of half slab h (columns 128 h ..) of this wave's record -> As[buf] rows 16 w ..
    auto expand = [&](int h, int buf, uint32_t d, const ChunkRegs& c0, const Seq& sq) {
        const int cb = h * GB_HS;
        uint32_t* base = reinterpret_cast<uint32_t*>(As + (size_t(buf) * GB_ROWS + wave * 16) * GB_ASTR) + lane;   // columns 2 lane, 2 lane + 1
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pos = r < 8 ? r + 8 : r - 8;
            const uint32_t hl = __builtin_amdgcn_readlane(hilo_lane, r);
            const uint32_t m = (d >> pos) & 0x00010001u;
            const uint32_t sel = m * 0x0202u + 0x01000100u;     // per half: bytes {1,0} (lo) or {3,2} (hi)
            base[r * (GB_ASTR / 2)] = __builtin_amdgcn_perm(hl, hl, sel);
        }
        asm volatile("" ::: "memory");                          // the overlays below must follow the plane (other store types)
        uint16_t* arow = reinterpret_cast<uint16_t*>(As + (size_t(buf) * GB_ROWS + wave * 16 + rho_s) * GB_ASTR);
        scatter(c0, cb, arow);                               // pass 0 (4 chunks per row and slab) was loaded a half slab ahead
        {
            const int n = sq.fn + sq.tn;
            for (int q = slot + 4; __any(q < n); q += 4) scatter(load_chunk(q, sq), cb, arow);
        }
        asm volatile("" ::: "memory");
        for (int k = lane; k < nexc; k += GW) {                 // explicit values last
            const uint2 ex = exc[k];
            const uint32_t col = (ex.x & 0xFFFFu) - uint32_t(cb);
            if (col < uint32_t(GB_HS))
                reinterpret_cast<uint16_t*>(As + (size_t(buf) * GB_ROWS + wave * 16 + (ex.x >> 16)) * GB_ASTR)[col] =
                    __builtin_bit_cast(uint16_t, _Float16(__builtin_bit_cast(float, ex.y)));
        }
        asm volatile("" ::: "memory");
    };

    // ---- x staging: thread -> 4 x (token, 16-byte unit) of a 64-column sub-step ---------------------------------------
    // Two register stages: x of sub-step u+2 is requested before the MFMAs of sub-step u, x of u+1 (requested one
    // sub-step earlier) is written to LDS after them -- every load has two sub-steps to arrive (one workgroup per CU: only
    // the partner wave on the SIMD hides latency otherwise).
    struct XRegs { u32x4 v[4]; };
    auto load_x = [&](int u, XRegs& xr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + j * (NWAVE * GW), tok = tok0 + (idx >> 3), col = u * GB_XC + (idx & 7) * 8;
            u32x4 v = {0, 0, 0, 0};
            if (tok < M && col < K) v = *reinterpret_cast<const u32x4*>(a.x + size_t(tok) * K + col);
            xr.v[j] = v;
        }
    };
    auto store_x = [&](int buf, const XRegs& xr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + j * (NWAVE * GW), t = idx >> 3, un = (idx & 7) ^ ((t >> 1) & 7);
            *reinterpret_cast<u32x4*>(Xs + (size_t(buf) * GB_TOK + t) * GB_XC + un * 8) = xr.v[j];
        }
    };

    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    // one 64-column sub-step: 2 k-steps x (4 A + 4 B fragments, 16 MFMAs)
    auto mfma_step = [&](int abuf, int colbase, int xbuf) {
        const _Float16* ap = As + (size_t(abuf) * GB_ROWS + wr * 64 + row_a) * GB_ASTR + colbase + kblk * 8;
        const _Float16* xp = Xs + (size_t(xbuf) * GB_TOK + wc * 64 + row_a) * GB_XC;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8h af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const v8h*>(ap + size_t(i) * 16 * GB_ASTR + ks * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // token t = wc*64 + 16 j + row_a: (t >> 1) & 7 == ((16 j + row_a) >> 1) & 7 == (row_a >> 1) & 7
                const int un = (ks * 4 + kblk) ^ ((row_a >> 1) & 7);
                bf[j] = *reinterpret_cast<const v8h*>(xp + size_t(j) * 16 * GB_XC + un * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- prologue: half slab 0 expanded, x sub-step 0 staged ----------------------------------------------------------
    // Half slab h+1 is expanded while half slab h is multiplied.  cA / sq: chunk data and ranges of the slab the NEXT
    // expansion belongs to; they are replaced right after that slab's second half has been expanded, a full iteration
    // before their next use.  dcur: the sign-plane dword of the next expansion, dnxt: of the one after.
    const uint32_t* tile_dw = reinterpret_cast<const uint32_t*>(rec + tiles_off) + lane * 4;
    auto load_dw = [&](int h) -> uint32_t {              // dword (h & 3) of panel h >> 2 for this lane
        return (h >> 2) < P ? __builtin_nontemporal_load(tile_dw + size_t(h >> 2) * 256 + (h & 3)) : 0u;
    };
    const int NH = 2 * NS, NU = 4 * NS;                  // half slabs, 64-column sub-steps
    uint32_t e0 = tab(0), e1 = tab(1), e2 = tab(2);
    Seq sq = seq_of(0u, e0);
    ChunkRegs cA = load_chunk(slot, sq);
    XRegs xa, xb;
    load_x(0, xa);
    if (1 < NU) load_x(1, xb);
    uint32_t dcur = load_dw(0), dnxt = load_dw(1);
    expand(0, 0, dcur, cA, sq);
    dcur = dnxt; dnxt = load_dw(2);
    store_x(0, xa);
    __syncthreads();

    // sub-step u: request x(u+2), multiply, publish x(u+1).  `xa` holds x(u+1) on even u, `xb` on odd u (NU is even).
    auto substep = [&](int u, int abuf, int colbase, XRegs& cur, XRegs& nxt) {
        if (!(PBL_GEMM_ABLATE & 2) && u + 2 < NU) load_x(u + 2, nxt);   // nxt's previous content (x(u)) went to LDS one sub-step ago
        if (!(PBL_GEMM_ABLATE & 4)) mfma_step(abuf, colbase, u & 1);
        if (!(PBL_GEMM_ABLATE & 2) && u + 1 < NU) store_x((u + 1) & 1, cur);
    };
    // after the expansion of half slab hn: fetch what the one after needs
    auto after_expand = [&](int hn) {
        if (hn & 1) {                                    // hn was a slab's second half: its chunks are done with
            const int sn = (hn >> 1) + 1;
            if (sn < NS) {
                sq = seq_of(e0, e1);
                cA = load_chunk(slot, sq);
            }
            e0 = e1; e1 = e2; e2 = tab(sn + 2);
        }
        dcur = dnxt;
        dnxt = load_dw(hn + 2);
    };
    for (int h = 0; h < NH; ++h) {
        const int hn = h + 1;                            // the half slab to expand now
        const bool more = hn < NH;
        if (!(PBL_GEMM_ABLATE & 1) && early && more) { expand(hn, hn & 1, dcur, cA, sq); after_expand(hn); }
        substep(2 * h, h & 1, 0, xb, xa);                // even sub-step: x(u+1) is in xb, x(u+2) goes to xa
        if (!(PBL_GEMM_ABLATE & 8)) __syncthreads();
        if (!(PBL_GEMM_ABLATE & 1) && !early && more) { expand(hn, hn & 1, dcur, cA, sq); after_expand(hn); }
        substep(2 * h + 1, h & 1, GB_XC, xa, xb);
        if (!(PBL_GEMM_ABLATE & 8)) __syncthreads();
    }

    // ---- epilogue: accumulators -> Ys[token][row] (fp16, + bias) in LDS -> contiguous stores ----------------------------
    _Float16* Ys = reinterpret_cast<_Float16*>(smem_b);                  // [256][GB_ASTR]: the A tiles are dead
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rloc = wr * 64 + i * 16 + 4 * kblk;                    // 4 consecutive rows held by this lane
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (L.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t row = rowblk * GB_ROWS + rloc + r;
                b4[r] = row < L.N ? L.bias[row] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = wc * 64 + j * 16 + row_a;
            _Float16 h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = _Float16(acc[i][j][r] + b4[r]);
            uint2 pk;
            pk.x = uint32_t(__builtin_bit_cast(uint16_t, h[0])) | (uint32_t(__builtin_bit_cast(uint16_t, h[1])) << 16);
            pk.y = uint32_t(__builtin_bit_cast(uint16_t, h[2])) | (uint32_t(__builtin_bit_cast(uint16_t, h[3])) << 16);
            *reinterpret_cast<uint2*>(Ys + size_t(t) * GB_ASTR + rloc) = pk;
        }
    }
    __syncthreads();
    const uint32_t row0 = rowblk * GB_ROWS;
    const bool vec = (L.N & 7) == 0 && row0 + GB_ROWS <= L.N;            // whole 16-byte units, all rows exist
    for (int idx = tid; idx < GB_TOK * (GB_ROWS / 8); idx += NWAVE * GW) {
        const int t = idx >> 4, un = idx & 15, tok = tok0 + t;
        if (tok >= M) continue;
        _Float16* dst = a.y + size_t(tok) * L.N + row0 + un * 8;
        const _Float16* src = Ys + size_t(t) * GB_ASTR + un * 8;
        if (vec) *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
        else
            for (int e = 0; e < 8; ++e)
                if (row0 + un * 8 + e < L.N) dst[e] = src[e];
    }
}

}  // namespace

extern "C" int pbl_gemm_f16(const pbl_layer* layer, const void* x, void* y, int M, void* stream) {
    if (!layer || !layer->blob || !x || !y || M < 1) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(layer->blob) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
        return PBL_ERR_MISALIGNED;
    // exact only when every weight is an fp16 number: layers packed from an fp16 checkpoint
    if (layer->G != 1 || (layer->K & 7) || !(layer->flags & PBL_FLAG_SAL_F16) || !(layer->flags & PBL_FLAG_SLABS) ||
        !(layer->flags & PBL_FLAG_TAIL_REPEAT))
        return PBL_ERR_UNSUPPORTED;
    GemmArgs a;
    a.L = *layer; a.x = static_cast<const _Float16*>(x); a.y = static_cast<_Float16*>(y); a.M = M;
    const void* k = reinterpret_cast<const void*>(pbl_gemm_kernel);
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(GB_LDS)) != hipSuccess) return PBL_ERR_LAUNCH;
    void* argv[] = {&a};
    const dim3 grid(((layer->NRB + NWAVE - 1) / NWAVE) * uint32_t((M + GB_TOK - 1) / GB_TOK));
    return hipLaunchKernel(k, grid, dim3(NWAVE * GW), argv, GB_LDS, static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}
