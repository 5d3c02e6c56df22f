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
// Workgroup = 16 waves, SPECIALISED: 8 consumer waves that only read LDS and issue MFMAs, 8 producer waves that only
// fetch and expand.  (A first version let every wave expand its own record between its MFMAs: 545 TFLOP/s -- every wave
// waited half of the time, profiles/r02b, while the MFMA loop by itself ran at 1.24 PFLOP/s; so that loop gets waves of
// its own.  4 producers could not keep up: ~450 instructions per sub-step each.)
// Tile: 8 records (128 output rows) x 256 tokens; K is walked in half slabs of 128 columns = 2 sub-steps of 64.
//   A operand  As[2][128 rows][128 + 8] fp16 (double buffered).  Producer p expands record p of the NEXT half slab: the
//              sign plane in the even sub-step, the salients in the odd one.  Sign plane -- per (row, dword) shift / and /
//              mad build a v_perm_b32 selector
//              that picks {hi, lo} for two columns, one ds_write_b32 stores them; salient chunks of the slab through the
//              packer's slab index, FOUR LANES PER CHUNK (4 entries each: short dependent chains, no idle lanes at low
//              density; the lane's first column is col0 + a v_sad_u8 byte sum of the preceding deltas); exceptions last.
//   B operand  Xs[2][256 tokens][64] fp16, XOR-swizzled 16-byte units (conflict-free b128 fragment reads without
//              padding): the 512 producer threads keep sub-step u+1 in registers (requested a sub-step earlier), write it
//              at the start of sub-step u and request u+2.
//   MFMA       consumers as 2 (rows) x 4 (tokens): a wave owns 64 rows x 64 tokens = 16 accumulator tiles; per 32-column
//              k-step 4 A + 4 B fragment reads feed 16 MFMAs.  No global memory access in their loop.
//   Sync       one workgroup barrier per sub-step; a buffer is written in the sub-step(s) after its last readers have
//              passed a barrier and read after the writers have passed the next one.
//   Epilogue   accumulators -> LDS [token][row] -> contiguous 16-byte stores by all 1024 threads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GW 64
#define NCONS 8                   // consumer (MFMA) waves
#define NPROD 8                   // producer (expand + stage) waves, one record each
#define NREC 8                    // records per workgroup tile
#define GB_ROWS (NREC * 16)
#define GB_TOK 256
#define GB_HS 128                 // columns per As buffer (half a slab)
#define GB_ASTR (GB_HS + 8)       // halves per As row: 272 B, the 16 row-lanes of a b128 read hit distinct banks
#define GB_XC 64                  // columns per x sub-step
#define GB_AS_BYTES (size_t(GB_ROWS) * GB_ASTR * 2)
#define GB_XS_BYTES (size_t(GB_TOK) * GB_XC * 2)
#define GB_LDS (2 * GB_AS_BYTES + 2 * GB_XS_BYTES)
// performance-analysis hook (tools/build_variant.sh): bit 0 no expansion in the loop, 1 no x staging in the loop, 2 no MFMA,
// 3 no sign-plane expansion, 4 no salient overlay.
// 0 in every shipped build (results are wrong otherwise).
#ifndef PBL_GEMM_ABLATE
#define PBL_GEMM_ABLATE 0
#endif

namespace {

__device__ __forceinline__ _Float16 round_f16_twice(float prod) {
    asm volatile("" : "+v"(prod));   // keep the fp32 product: the checkpoint value is double rounded
    return _Float16(prod);
}

struct GemmArgs {
    pbl_layer L;
    const _Float16* x;      // [M, K]
    _Float16* y;            // [M, N]
    int M;
};

struct Seq { int fb, fn, tb, tn; };            // the row's full / tail chunks that overlap a slab: first index, count

// a quarter of a salient chunk as one lane holds it: all 16 byte steps (for the prefix), its 4 codes, the first column
struct ChunkQ {
    u32x4 d4;
    uint32_t q;
    int col0;               // < 0: no chunk
};

// Everything a producer wave keeps per record it expands (two of these per wave).
struct Rec {
    const uint8_t* rec;     // record base
    const uint16_t* col0p;
    const u32x4* deltap;
    const uint32_t* codew;  // codes as dwords: chunk c, quarter s at [4 c + s]
    const uint2* exc;
    const uint32_t* tabrow; // slab-index row of this lane's row
    const uint32_t* tile_dw;
    int nfull, nexc, NS, P;
    pbl_rowinfo ri;         // this lane's row (lane >> 2)
    float ss, sz;           // ... and its code grid
    uint32_t hilo_lane;     // lane (r & 15): fp16 {hi : lo} of row r
    uint32_t e0, e1, e2;    // slab-index entries of slabs s, s+1, s+2 (s = the slab of the next expansion)
    Seq sq;
    ChunkQ c0, c1;          // rounds 0 and 1 of the slab of the next expansion
    uint32_t dcur, dnxt;    // sign-plane dwords of the next expansion and the one after
};

__device__ __forceinline__ uint32_t tab_at(const Rec& R, int s) { return (s >= 0 && s < R.NS) ? R.tabrow[s] : 0u; }
__device__ __forceinline__ Seq seq_of(uint32_t pe, uint32_t e) {
    Seq q;
    q.fb = int(PBL_SLAB_FE(pe)) - int(PBL_SLAB_FBACK(e)); q.fn = int(PBL_SLAB_FE(e)) - q.fb;
    q.tb = int(PBL_SLAB_TE(pe)) - int(PBL_SLAB_TBACK(e)); q.tn = int(PBL_SLAB_TE(e)) - q.tb;
    return q;
}
// round j of the row's slab sequence (its full chunks, then its tail chunks): the same chunk for the 4 lanes of a row
__device__ __forceinline__ ChunkQ load_chunkq(const Rec& R, int j, const Seq& sq, int sub) {
    ChunkQ r;
    r.col0 = -1; r.d4 = u32x4{0, 0, 0, 0}; r.q = 0;
    int c = -1;
    if (j < sq.fn) c = int(R.ri.start) + sq.fb + j;
    else if (j - sq.fn < sq.tn) c = R.nfull + int(R.ri.tailidx) + sq.tb + (j - sq.fn);
    if (c >= 0) { r.d4 = R.deltap[c]; r.q = R.codew[4 * c + sub]; r.col0 = int(R.col0p[c]); }
    return r;
}
__device__ __forceinline__ uint32_t load_dw(const Rec& R, int h) {      // dword (h & 3) of panel h >> 2 for this lane
    return (h >> 2) < R.P ? __builtin_nontemporal_load(R.tile_dw + size_t(h >> 2) * 256 + (h & 3)) : 0u;
}

// entries 4 sub .. 4 sub + 3 of one chunk -> the tile row (byte address arow_b), half slab starting at column cb.
// Tail padding repeats the last entry (PBL_FLAG_TAIL_REPEAT), so a writer needs no count.
__device__ __forceinline__ void scatter_q(const ChunkQ& c, int cb, int sub, uint32_t pad_b, char* arow_b, float ss, float sz) {
    if (c.col0 < 0) return;
    // byte sum of the deltas that precede this quarter (deltas are stored doubled = byte steps in an fp16 row)
    uint32_t pre = 0;
    pre = sub > 0 ? __builtin_amdgcn_sad_u8(c.d4[0], 0u, pre) : pre;
    pre = sub > 1 ? __builtin_amdgcn_sad_u8(c.d4[1], 0u, pre) : pre;
    pre = sub > 2 ? __builtin_amdgcn_sad_u8(c.d4[2], 0u, pre) : pre;
    const uint32_t dd = sub == 0 ? c.d4[0] : (sub == 1 ? c.d4[1] : (sub == 2 ? c.d4[2] : c.d4[3]));
    uint32_t off = uint32_t(2 * (c.col0 - cb)) + pre;          // byte offset in the row; wraps for entries left of the slab
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        off += (dd >> (8 * e)) & 0xFFu;
        const uint32_t o = min(off, pad_b);                     // outside the half slab: one of the row's pad columns
        const float qf = float((c.q >> (8 * e)) & 0xFFu);
        *reinterpret_cast<uint16_t*>(arow_b + o) = __builtin_bit_cast(uint16_t, round_f16_twice(ss * (qf - sz)));
    }
}

// sign plane of half slab h (columns 128 h ..) of record R -> rows 16 slot .. of As[buf]: every column of the 16 rows
__device__ __forceinline__ void expand_sign(Rec& R, const uint32_t (&hl)[16], _Float16* As, int buf, int slot, int lane) {
    const uint32_t d = R.dcur;
    uint32_t* base = reinterpret_cast<uint32_t*>(As + (size_t(buf) * GB_ROWS + slot * 16) * GB_ASTR) + lane;   // columns 2 lane, 2 lane + 1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pos = r < 8 ? r + 8 : r - 8;
        const uint32_t m = (d >> pos) & 0x00010001u;
        const uint32_t sel = m * 0x0202u + 0x01000100u;         // per half: bytes {1,0} (lo) or {3,2} (hi)
        base[r * (GB_ASTR / 2)] = __builtin_amdgcn_perm(hl[r], hl[r], sel);
    }
    asm volatile("" ::: "memory");
}

// Loads for the expansion of half slab hn and later, issued at the START of the even sub-step (right after x went to LDS
// and BEFORE the next x request): vmcnt retires in order and hipcc waits vmcnt(0) in front of the LDS store of x, so a load
// issued after an x request would put its whole latency into the next sub-step's critical path.
__device__ __forceinline__ void prefetch_for(Rec& R, int hn, int sub) {
    R.dcur = R.dnxt;                                            // (landed a half slab ago)
    R.dnxt = load_dw(R, hn + 1);
    if (!(hn & 1)) {                                            // hn opens slab hn / 2: its chunks (first used one sub-step from now)
        const int sn = hn >> 1;
        if (sn < R.NS) {
            R.sq = seq_of(R.e0, R.e1);
            R.c0 = load_chunkq(R, 0, R.sq, sub);
            R.c1 = load_chunkq(R, 1, R.sq, sub);
        }
        R.e0 = R.e1; R.e1 = R.e2; R.e2 = tab_at(R, sn + 2);
    }
}

// salients + exceptions of half slab h over the plane written one sub-step earlier
__device__ __forceinline__ void expand_sal(Rec& R, int h, _Float16* As, int buf, int slot, int lane) {
    const int cb = h * GB_HS, rho = lane >> 2, sub = lane & 3;
    char* arow_b = reinterpret_cast<char*>(As + (size_t(buf) * GB_ROWS + slot * 16 + rho) * GB_ASTR);
    const uint32_t pad_b = uint32_t(2 * (GB_HS + (lane & 7)));
    scatter_q(R.c0, cb, sub, pad_b, arow_b, R.ss, R.sz);        // rounds 0 and 1 were loaded a half slab ahead
    scatter_q(R.c1, cb, sub, pad_b, arow_b, R.ss, R.sz);
    {
        const int n = R.sq.fn + R.sq.tn;
        for (int j = 2; __any(j < n); ++j) scatter_q(load_chunkq(R, j, R.sq, sub), cb, sub, pad_b, arow_b, R.ss, R.sz);
    }
    asm volatile("" ::: "memory");
    for (int k = lane; k < R.nexc; k += GW) {                   // explicit values last
        const uint2 ex = R.exc[k];
        const uint32_t col = (ex.x & 0xFFFFu) - uint32_t(cb);
        if (col < uint32_t(GB_HS))
            reinterpret_cast<uint16_t*>(As + (size_t(buf) * GB_ROWS + slot * 16 + (ex.x >> 16)) * GB_ASTR)[col] =
                __builtin_bit_cast(uint16_t, _Float16(__builtin_bit_cast(float, ex.y)));
    }
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void init_rec(Rec& R, const pbl_layer& L, uint32_t rb, int lane) {
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    R.rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    R.nfull = __builtin_amdgcn_readfirstlane(info.y);
    const int ntail = __builtin_amdgcn_readfirstlane(info.z);
    R.nexc = __builtin_amdgcn_readfirstlane(info.w);
    const uint32_t nchu = uint32_t(R.nfull + ntail);
    R.P = int(L.P);
    R.NS = int((L.K + PBL_SLAB_COLS - 1) / PBL_SLAB_COLS);
    const uint32_t tiles_off = PBL_TILES_OFF(1u);
    const uint8_t* sal = R.rec + tiles_off + uint32_t(R.P) * 1024u;
    R.col0p = reinterpret_cast<const uint16_t*>(sal);
    R.deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    R.codew = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_CODE_OFF(nchu));
    R.exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), true));
    const uint32_t* slabtab = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_SLAB_OFF(nchu, uint32_t(ntail), uint32_t(R.nexc), true));
    R.tabrow = slabtab + (lane >> 2) * R.NS;
    R.tile_dw = reinterpret_cast<const uint32_t*>(R.rec + tiles_off) + lane * 4;
    const pbl_rowparams* params = reinterpret_cast<const pbl_rowparams*>(R.rec + PBL_REC_PARAMS_OFF);
    R.ri = reinterpret_cast<const pbl_rowinfo*>(R.rec + PBL_REC_ROWINFO_OFF)[lane >> 2];
    const pbl_rowparams ps = params[lane >> 2];
    R.ss = ps.sscale; R.sz = ps.szero;
    const pbl_rowparams pr = params[lane & 15];
    const uint32_t hh = __builtin_bit_cast(uint16_t, _Float16(pr.hi)), ll = __builtin_bit_cast(uint16_t, _Float16(pr.lo));
    R.hilo_lane = (hh << 16) | ll;
    R.e0 = tab_at(R, 0); R.e1 = tab_at(R, 1); R.e2 = tab_at(R, 2);
    R.sq = seq_of(0u, R.e0);
    R.c0 = load_chunkq(R, 0, R.sq, lane & 3);
    R.c1 = load_chunkq(R, 1, R.sq, lane & 3);
    R.dcur = load_dw(R, 0);
    R.dnxt = load_dw(R, 1);
}

__global__ __launch_bounds__((NCONS + NPROD) * GW) void pbl_gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const pbl_layer& L = a.L;
    const int K = int(L.K), M = a.M;
    const int NS = (K + PBL_SLAB_COLS - 1) / PBL_SLAB_COLS;
    const int NH = 2 * NS, NU = 4 * NS;                  // half slabs, 64-column sub-steps
    // XCD-aware work order (speed only; measured neutral so far): workgroup b runs on XCD b % 8; every XCD gets a CONTIGUOUS
    // range of the token-tile-major work list, so the workgroups resident on an XCD share one 256-token slab of x.
    const uint32_t nrbk = (L.NRB + NREC - 1) / NREC, nwg = gridDim.x;
    const uint32_t xq = nwg >> 3, xr_ = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const uint32_t wg = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + xi;
    const uint32_t rowblk = wg % nrbk;
    const int tok0 = int(wg / nrbk) * GB_TOK;

    _Float16* As = reinterpret_cast<_Float16*>(smem_b);                              // [2][128][GB_ASTR]
    _Float16* Xs = reinterpret_cast<_Float16*>(smem_b + 2 * GB_AS_BYTES);            // [2][256][64], 16-byte units XOR (token >> 1) & 7

    v4f acc[4][4];
    const int row_a = lane & 15, kblk = lane >> 4;       // fragment coordinates
    const int wr = (wave >> 2) & 1, wc = wave & 3;       // consumer wave's 64-row x 64-token block

    if (wave >= NCONS) {
        // =================================== producer waves =========================================================
        const int p = wave - NCONS, pt = tid - NCONS * GW;                         // producer index = record slot, producer thread 0..511
        Rec R;
        init_rec(R, L, min(rowblk * NREC + p, L.NRB - 1), lane);   // (a record beyond the layer mirrors the last one; its rows are never stored)
        uint32_t hl[16];                                 // the record's 16 level pairs, wave uniform: 16 SGPRs for the whole kernel
#pragma unroll
        for (int r = 0; r < 16; ++r) hl[r] = __builtin_amdgcn_readlane(R.hilo_lane, r);
        // x staging: producer thread -> 4 x (token, 16-byte unit) of a 64-column sub-step: tokens (pt >> 3) + 64 j, unit pt & 7
        const int xtok = pt >> 3, xun = pt & 7;
        const _Float16* xsrc = a.x + size_t(tok0 + xtok) * K + xun * 8;
        const size_t xjs = size_t(64) * K;
        _Float16* xdst = Xs + size_t(xtok) * GB_XC + (xun ^ ((xtok >> 1) & 7)) * 8;     // (token + 64 j) >> 1 & 7 == (token >> 1) & 7
        u32x4 xr[4];
        auto load_x = [&](int u) {
            const bool colok = u * GB_XC + xun * 8 < K;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 v = {0, 0, 0, 0};
                if (colok && tok0 + xtok + 64 * j < M) v = *reinterpret_cast<const u32x4*>(xsrc + j * xjs + size_t(u) * GB_XC);
                xr[j] = v;
            }
        };
        auto store_x = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(xdst + (size_t(buf) * GB_TOK + 64 * j) * GB_XC) = xr[j];
        };
        load_x(0);
        expand_sign(R, hl, As, 0, p, lane);
        expand_sal(R, 0, As, 0, p, lane);
        store_x(0);
        if (1 < NU) load_x(1);
        __syncthreads();
        for (int u = 0; u < NU; ++u) {
            const int hn = (u >> 1) + 1;
            const bool ex = !(PBL_GEMM_ABLATE & 1) && hn < NH;
            if (!(PBL_GEMM_ABLATE & 2) && u + 1 < NU) store_x((u + 1) & 1);   // x(u+1), requested a sub-step ago; its buffer was last read in sub-step u-1
            if (ex && !(u & 1)) prefetch_for(R, hn, lane & 3);                 // before the x request (see prefetch_for)
            if (!(PBL_GEMM_ABLATE & 2) && u + 2 < NU) load_x(u + 2);
            if (ex) {                                    // into the buffer last read in half slab h-1: plane first, salients a sub-step later
                if (u & 1) { if (!(PBL_GEMM_ABLATE & 16)) expand_sal(R, hn, As, hn & 1, p, lane); }
                else if (!(PBL_GEMM_ABLATE & 8)) expand_sign(R, hl, As, hn & 1, p, lane);
            }
            __syncthreads();
        }
    } else {
        // =================================== consumer waves =========================================================
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int u = 0; u < NU; ++u) {
            const int abuf = (u >> 1) & 1, colbase = (u & 1) * GB_XC, xbuf = u & 1;
            const _Float16* ap = As + (size_t(abuf) * GB_ROWS + wr * 64 + row_a) * GB_ASTR + colbase + kblk * 8;
            const _Float16* xp = Xs + (size_t(xbuf) * GB_TOK + wc * 64 + row_a) * GB_XC;
            if (!(PBL_GEMM_ABLATE & 4)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    v8h af[4], bf[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const v8h*>(ap + size_t(i) * 16 * GB_ASTR + ks * 32);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // token t = wc*64 + 16 j + row_a: (t >> 1) & 7 == (row_a >> 1) & 7
                        const int un = (ks * 4 + kblk) ^ ((row_a >> 1) & 7);
                        bf[j] = *reinterpret_cast<const v8h*>(xp + size_t(j) * 16 * GB_XC + un * 8);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: consumers' accumulators -> Ys[token][row] (fp16, + bias) in LDS -> contiguous stores by everybody ----
    _Float16* Ys = reinterpret_cast<_Float16*>(smem_b);                  // [256][GB_ASTR]: the A tiles are dead
    if (wave < NCONS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rloc = wr * 64 + i * 16 + 4 * kblk;                // 4 consecutive rows held by this lane
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
    }
    __syncthreads();
    const uint32_t row0 = rowblk * GB_ROWS;
    const bool vec = (L.N & 7) == 0 && row0 + GB_ROWS <= L.N;            // whole 16-byte units, all rows exist
    for (int idx = tid; idx < GB_TOK * (GB_ROWS / 8); idx += (NCONS + NPROD) * GW) {
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
    const dim3 grid(((layer->NRB + NREC - 1) / NREC) * uint32_t((M + GB_TOK - 1) / GB_TOK));
    return hipLaunchKernel(k, grid, dim3((NCONS + NPROD) * GW), argv, GB_LDS, static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}
