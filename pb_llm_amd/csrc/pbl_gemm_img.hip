// pbl_gemm_img.hip -- GEMM regime (prefill: more than 32 rows of x) over a per-layer GEMM IMAGE of the packed weight.
// Replaces F.linear(x, W_fq, b) over the dense fp16 fake-quant weight (gptq_pb/eval_ppl_utils.py:55-64, evaluate.py:126-145:
// the reference's perplexity loops call every nn.Linear with 2048 rows) without the dense weight ever existing in HBM.
//
// Round 4.  The round-3 kernel (pbl_gemm_big.hip) was measured from inside (tools/trace_gemm.py, profiles/r04_gemm.md): at
// 4096^2 x 2048 its loop takes 68 us against 36 - 38 us for the bare MFMA stream at the package power cap (1.85 - 2.0 GHz),
// and BOTH roles need that long -- the eight MFMA waves (fragment reads: +16 us over the bare stream; their own x staging
// by LDS-DMA: +10 us) and the four expanding waves (~330 instructions per half slab and wave: request addressing, range
// look-ups, masked stores, branches around every one of them).  What this kernel changes:
//   * the layer is re-laid ONCE (pbl_gemm_image_build, kept with the layer like round 3's salient list) into slots of 1 - 5 KiB,
//     one per (16-row record, 128-column half slab): per lane the sign-plane dword and EW - 1 ready-to-store salient words
//     {LDS offset : fp16 value}, padded with idempotent repeats -- so an expanding wave issues ONE 16-byte load per lane
//     and record from an address that only moves by a scalar add, and stores every word unconditionally: no ranges, no
//     clamps, no exec masking, no branches; the row levels are 16 SGPRs read with one scalar load.  ~105 instructions per
//     record and 64-column step instead of ~165 per record and half slab plus the request bookkeeping.
//   * four MFMA waves of 128 rows x 64 tokens (eight 32 x 32 accumulator tiles, 6 fragment reads per 8 MFMAs: 192 KB of LDS
//     reads per half slab instead of 320) that do NOTHING but read fragments and multiply;
//   * x is staged by the expanding waves (LDS-DMA, 8 KiB each per 64-column step) into a ring shared by the workgroup, one
//     workgroup barrier per 64-column step; the MFMA waves never issue a vector-memory instruction in the loop.
// Every weight still enters v_mfma_f32_32x32x16_f16 as the fp16 number a dense fp16 copy of the layer holds, and every
// accumulator sums its k-steps in the same order as pbl_gemm_big.hip: the two kernels agree bit for bit.
//
// Round 6 (template parameter XF, the module path's default).  Taken apart with PBL_IMG_ABLATE builds the loop above spends a
// quarter of a round staging x through LDS.  With x handed over as a FRAGMENT-MAJOR copy (pbl_x_to_fragments: one small kernel
// per distinct activation tensor) the MFMA waves load their B fragments straight into registers, the LDS carries A alone, the
// workgroup synchronises once per half slab, and the same bits come out 6 - 9 % sooner (profiles/r06_gemm.md: what bounds the
// kernel then -- the vector-memory path into registers --, and everything tried on top that did not pay).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/pbl.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define GW 64
#define GI_NCONS 4
#define GI_NPROD 4
#define GI_ROWS 128
#define GI_TOK 256
#define GI_HS 128                 // columns per A stage
#define GI_XC 64                  // columns per step (x slot)
#define GI_AS_STAGE (GI_ROWS * GI_HS * 2)         // 32768 B
#define GI_XSLOT (GI_TOK * GI_XC * 2)              // 32768 B
#define GI_X_OFF (2 * GI_AS_STAGE)
#define GI_LDS (GI_X_OFF + 3 * GI_XSLOT)           // 163840 B
#define GI_MAX_NH 127
#define GI_NVMAX 5                // 16-byte vectors per lane in the largest slot
#define GI_IMG_MAGIC 0x33494250u                   // "PBI3" (slots vector-major, sized per record and half slab)
#define GI_PREP_THREADS 1024
#ifndef PBL_GEMM_PPRIO
#define PBL_GEMM_PPRIO 1
#endif
// performance-analysis hook (tools/build_variant.sh): bit 0 no expansion in the loop, bit 1 no x staging in the loop, bit 2 no
// MFMA, bits 3 / 4 x staging that re-reads half / an eighth of its bytes (round 6).  0 in every shipped build (results are wrong otherwise).
#ifndef PBL_IMG_ABLATE
#define PBL_IMG_ABLATE 0
#endif
#ifndef PBL_XF_REQ_PROBE
#define PBL_XF_REQ_PROBE 0
#endif
#ifndef PBL_XF_BMOD
#define PBL_XF_BMOD ""                // XF: cache-policy bits of the B-fragment loads (A/B builds: " nt", " sc0", " sc1", " sc0 sc1")
#endif
#ifndef PBL_XF_DEPTH
#define PBL_XF_DEPTH 4                // XF: k-steps the MFMA waves request their B fragments ahead (4: one 64-column step, 8: two)
#endif

// timeline probe, as in pbl_gemm_big.hip (tools/trace_gemm.py)
#ifndef PBL_TRACE
#define PBL_TRACE 0
#endif
#if PBL_TRACE
static uint64_t* g_img_trace = nullptr;
extern "C" void pbl_debug_trace_gemm_img(void* p) { g_img_trace = static_cast<uint64_t*>(p); }
#define TR_STAMP(slot) do { if (tr && lane == 0) { tr[2 * (slot)] = __builtin_amdgcn_s_memrealtime(); tr[2 * (slot) + 1] = __builtin_readcyclecounter(); } } while (0)
#define TR_T0() const uint64_t tr_t0 = __builtin_readcyclecounter()
#define TR_ADD(acc) acc += __builtin_readcyclecounter() - tr_t0
#else
#define TR_STAMP(slot) do {} while (0)
#define TR_T0() do {} while (0)
#define TR_ADD(acc) do {} while (0)
#endif

namespace {

// The image: [header 64 B][rbase: NRB u32][rtab: NRB x 128 u32][slots][levels: NRB x G x 16 u32].
//   rbase[r]    where record r's slots start, in 256-byte units from slots_off;
//   rtab[r][h]  slot of half slab h of record r: (offset from the record's start, 256-byte units) | nv << 16, nv = 1 .. 5 sixteen-byte
//               vectors per lane = 3, 7, 11, 15 or 19 entry words per lane = up to 192 / 448 / 704 / 960 / 1216 entries.  Every slot is
//               sized for ITS OWN entry count (round 4, second layout): with one size per column (the maximum over the records) a
//               layer with 20 % salients needed 3 KiB for slots that hold 410 entries on average -- 106 MB where 71 do, and the
//               small-batch kernel is bound by exactly those bytes.
//   slot        [nv vectors][64 lanes][4] u32: word w of lane l at (w >> 2) * 256 + 4 l + (w & 3); word 0 the lane's sign-plane dword.
// Geometry words (host side, read back once after pbl_gemm_image_stats): geom[0] = all slots in 256-byte units, geom[1] = the largest nv.
struct ImgHeader {
    uint32_t magic, NH, NRB, G, K, N, flags, nvmax;
    uint64_t rtab_off, slots_off, levels_off, total;
};
static_assert(sizeof(ImgHeader) == 64, "image header is 64 bytes");
#define GI_TABW 128                // words of a record's slot table

__device__ __forceinline__ _Float16 round_f16_twice(float prod) {
    asm volatile("" : "+v"(prod));   // keep the fp32 product: an fp16-checkpoint value is double rounded
    return _Float16(prod);
}
__device__ __forceinline__ uint32_t h16(float v) { return uint32_t(__builtin_bit_cast(uint16_t, _Float16(v))); }

// ---- building the image ---------------------------------------------------------------------------------------------------
// One workgroup per record.  STATS: count the entries of every half slab of the record, size its slots (rtab[rb][..], rlen[rb] = the
// record's slots in 256-byte units, geom[1] = the largest nv; 0xFFFFFFFF: a slot would need more than five vectors).  Otherwise fill
// the record's slots (at rbase[rb], sized by rtab[rb]), copy its table row and start into the image, and write its level rows.
struct PrepArgs {
    uint8_t* img;                  // build: the image
    uint32_t* geom;                // stats: geom[1] (max nv)
    uint32_t* rlen;                // stats: out, slots of the record in 256-byte units; build: in, rbase (the exclusive prefix sum)
    uint32_t* rtab;                // stats: out; build: in
    uint64_t rtab_off, slots_off, levels_off, total;
    int resid;                     // build: the RESIDUAL image of an fp32-grid layer (see pbl_gemm_image_build_residual)
};
// the fp16 number an image holds for the fp32 value v: fp16(v), or -- residual image -- fp16(4096 (v - fp16(v))): what fp16 lost of
// v, scaled by 2^12 into fp16's normal range (exact up to ~2^-22 |v|)
__device__ __forceinline__ uint32_t img_h16(float v, int resid) {
    asm volatile("" : "+v"(v));      // keep the fp32 value: an fp16-checkpoint value is double rounded
    const _Float16 h = _Float16(v);
    if (!resid) return uint32_t(__builtin_bit_cast(uint16_t, h));
    return uint32_t(__builtin_bit_cast(uint16_t, _Float16(4096.f * (v - float(h)))));
}
template <bool STATS>
__global__ __launch_bounds__(GI_PREP_THREADS) void img_prep_kernel(pbl_layer L, PrepArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem_p);            // entries per half slab
    uint32_t* s_first = s_cnt + (GI_MAX_NH + 1);                       // the first word written to a half slab (padding repeats it)
    uint32_t* s_tab = s_first + (GI_MAX_NH + 1);                       // the record's slot table
    float* s_ss = reinterpret_cast<float*>(s_tab + GI_TABW);
    float* s_sz = s_ss + 16;
    pbl_rowinfo* s_ri = reinterpret_cast<pbl_rowinfo*>(s_sz + 16);
    uint8_t* s_crow = reinterpret_cast<uint8_t*>(s_ri + 16);
    const uint32_t rb = blockIdx.x;
    const int tid = threadIdx.x;
    const int NH = int((L.K + GI_HS - 1) / GI_HS);
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    const uint8_t* rec = blob + size_t(info.x) * 16;
    const int nfull = int(info.y), ntail = int(info.z), nexc = int(info.w), nch = nfull + ntail;
    const uint32_t nchu = uint32_t(nch);
    const bool has_crow = (L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16)) != 0;
    const uint32_t tiles_off = PBL_TILES_OFF(L.G);
    const uint8_t* sal = rec + tiles_off + L.P * 1024u;
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const uint32_t* codew = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_CODE_OFF(nchu));
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));
    const pbl_rowparams* params = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
    const float2* ghl = reinterpret_cast<const float2*>(rec + PBL_REC_GHL_OFF);
    const uint32_t* tile_dw = reinterpret_cast<const uint32_t*>(rec + tiles_off);
    uint8_t* const img = pa.img;
    uint8_t* slotrow = STATS ? nullptr : img + pa.slots_off + size_t(pa.rlen[rb]) * 256;
    auto slot_of = [&](uint32_t h) -> uint32_t* { return reinterpret_cast<uint32_t*>(slotrow + size_t(s_tab[h] & 0xFFFFu) * 256); };
    if (!STATS)
        for (int h = tid; h < GI_TABW; h += GI_PREP_THREADS) s_tab[h] = pa.rtab[size_t(rb) * GI_TABW + h];
    if (tid < 16) {
        s_ri[tid] = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF)[tid];
        s_ss[tid] = params[tid].sscale; s_sz[tid] = params[tid].szero;
    }
    for (int h = tid; h <= NH; h += GI_PREP_THREADS) { s_cnt[h] = 0; s_first[h] = 0; }
    __syncthreads();
    for (int r = 0; r < 16; ++r) {                           // chunk -> row (rowinfo: full chunks [start, +nfull), tails [tailidx, +ntail))
        const pbl_rowinfo ri = s_ri[r];
        for (int k = tid; k < int(ri.nfull) + int(ri.ntail); k += GI_PREP_THREADS)
            s_crow[k < int(ri.nfull) ? int(ri.start) + k : nfull + int(ri.tailidx) + (k - int(ri.nfull))] = uint8_t(r);
    }
    __syncthreads();
    auto put = [&](uint32_t h, uint32_t j, uint32_t word) {   // entry j of half slab h: lane j & 63, word 1 + (j >> 6)
        if (j == 0) s_first[h] = word;
        const uint32_t EW = 4u * (s_tab[h] >> 16);
        const uint32_t w = 1u + (j >> 6);
        if (j < 64u * (EW - 1u)) slot_of(h)[(w >> 2) * 256u + (j & 63u) * 4u + (w & 3u)] = word;
    };
    // a quarter of a chunk per thread
    for (int u = tid; u < 4 * nch; u += GI_PREP_THREADS) {
        const int c = u >> 2, sub = u & 3;
        const u32x4 dv = deltap[c];
        const uint32_t q = codew[u], cc = col0p[c];
        const uint32_t row = s_crow[c];
        uint32_t pre = 0;
        pre = sub > 0 ? __builtin_amdgcn_sad_u8(dv[0], 0u, pre) : pre;
        pre = sub > 1 ? __builtin_amdgcn_sad_u8(dv[1], 0u, pre) : pre;
        pre = sub > 2 ? __builtin_amdgcn_sad_u8(dv[2], 0u, pre) : pre;
        const uint32_t dd = sub == 0 ? dv[0] : (sub == 1 ? dv[1] : (sub == 2 ? dv[2] : dv[3]));
        uint32_t off[4];
        off[0] = 2u * cc + pre + (dd & 0xFFu);               // byte offsets in the fp16 row
        off[1] = off[0] + ((dd >> 8) & 0xFFu); off[2] = off[1] + ((dd >> 16) & 0xFFu); off[3] = off[2] + (dd >> 24);
        const float ss = s_ss[row], sz = s_sz[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // the padding of a tail chunk repeats its last entry with step 0 (PBL_FLAG_TAIL_REPEAT): not an entry of its own (real
            // columns rise strictly; only a chunk's FIRST entry may have step 0) -- counting it would size the slots of layers with
            // many short chunks (hessian salients) for up to 16 x their content
            if ((4 * sub + e) > 0 && ((dd >> (8 * e)) & 0xFFu) == 0u) continue;
            const uint32_t h = off[e] >> 8;
            const uint32_t pos = atomicAdd(&s_cnt[h], 1u);
            if (!STATS) {
                const float qf = float((q >> (8 * e)) & 0xFFu);
                const uint32_t v = img_h16(ss * (qf - sz), pa.resid);
                put(h, pos, ((row * 256u + ((off[e] & 0xFFu) ^ (row << 4))) << 16) | v);
            }
        }
    }
    for (int k = tid; k < nexc; k += GI_PREP_THREADS) {
        const uint2 ex = exc[k];
        const uint32_t col = ex.x & 0xFFFFu, row = ex.x >> 16, h = col >> 7, o = (2u * col) & 0xFFu;
        const uint32_t pos = atomicAdd(&s_cnt[h], 1u);
        if (!STATS) {
            const uint32_t v = img_h16(__builtin_bit_cast(float, ex.y), pa.resid);
            put(h, pos, ((row * 256u + (o ^ (row << 4))) << 16) | v);
        }
    }
    __syncthreads();
    if (STATS) {
        if (tid == 0) {
            uint32_t off = 0, nvmax = 0;
            for (int h = 0; h < GI_TABW; ++h) {
                uint32_t t = 0;
                if (h < NH) {
                    const uint32_t m = s_cnt[h];
                    const uint32_t nv = m <= 192u ? 1u : (m <= 448u ? 2u : (m <= 704u ? 3u : (m <= 960u ? 4u : (m <= 1216u ? 5u : 0u))));
                    nvmax = max(nvmax, nv ? nv : 0xFFFFFFFFu);
                    t = off | (nv << 16);
                    off += 4u * nv;                        // 1 KiB = four 256-byte units per vector
                }
                pa.rtab[size_t(rb) * GI_TABW + h] = t;
            }
            pa.rlen[rb] = off;
            atomicMax(pa.geom + 1, nvmax);
        }
        return;
    }
    // the plane dword of every (half slab, lane) and the padding: unused words repeat the half slab's first entry (the same
    // value to the same place: idempotent); a half slab without any entry rewrites position (row 0, column 0) with the value
    // the plane gives it
    for (int it = tid; it < NH * 64; it += GI_PREP_THREADS) {
        const int h = it >> 6, l = it & 63;
        const uint32_t d = uint32_t(h >> 2) < L.P ? tile_dw[size_t(h >> 2) * 256 + l * 4 + (h & 3)] : 0u;
        const uint32_t EW = 4u * (s_tab[h] >> 16);
        uint32_t* sl = slot_of(uint32_t(h)) + size_t(l) * 4;       // word k of this lane: sl[(k >> 2) * 256 + (k & 3)]
        sl[0] = d;
        const uint32_t n = s_cnt[h];
        uint32_t padw = s_first[h];
        if (n == 0) {
            const uint32_t d0 = uint32_t(h >> 2) < L.P ? tile_dw[size_t(h >> 2) * 256 + (h & 3)] : 0u;   // lane 0: columns 0, 1 of the half slab
            float hi, lo;
            if (L.G > 1) { const float2 v = ghl[(uint32_t(h) * GI_HS) / (L.K / L.G)]; hi = v.x; lo = v.y; }   // row 0's levels of that group
            else { hi = params[0].hi; lo = params[0].lo; }
            padw = ((d0 >> 8) & 1u) ? img_h16(hi, pa.resid) : img_h16(lo, pa.resid);                         // row 0 <-> bit 8 (pbl.h), offset 0
        }
        for (uint32_t k = 1; k < EW; ++k)
            if ((k - 1u) * 64u + uint32_t(l) >= n) sl[(k >> 2) * 256u + (k & 3u)] = padw;
    }
    // level rows: (hi - lo) mod 2^16 : lo as fp16 bit patterns, 16 per (record, group)
    uint32_t* lev = reinterpret_cast<uint32_t*>(img + pa.levels_off) + size_t(rb) * L.G * 16;
    for (uint32_t it = tid; it < 16u * L.G; it += GI_PREP_THREADS) {
        const uint32_t g = it >> 4, r = it & 15u;
        float hi, lo;
        if (L.G > 1) { const float2 v = ghl[size_t(r) * L.G + g]; hi = v.x; lo = v.y; }
        else { hi = params[r].hi; lo = params[r].lo; }
        const uint32_t hh = img_h16(hi, pa.resid), ll = img_h16(lo, pa.resid);
        lev[it] = (((hh - ll) & 0xFFFFu) << 16) | ll;
    }
    // the record's table row and start, into the image
    for (int h = tid; h < GI_TABW; h += GI_PREP_THREADS) reinterpret_cast<uint32_t*>(img + pa.rtab_off)[size_t(rb) * GI_TABW + h] = s_tab[h];
    if (tid == 0) reinterpret_cast<uint32_t*>(img + sizeof(ImgHeader))[rb] = pa.rlen[rb];
    if (rb == 0 && tid == 0) {
        ImgHeader* w = reinterpret_cast<ImgHeader*>(img);
        w->magic = GI_IMG_MAGIC; w->NH = uint32_t(NH); w->NRB = L.NRB; w->G = L.G; w->K = L.K; w->N = L.N; w->flags = L.flags; w->nvmax = 0;
        w->rtab_off = pa.rtab_off; w->slots_off = pa.slots_off; w->levels_off = pa.levels_off; w->total = pa.total;
    }
}

// rlen[0 .. NRB) -> its exclusive prefix sum in place; geom[0] = the sum.  One workgroup.
__global__ __launch_bounds__(1024) void img_scan_kernel(uint32_t* __restrict__ geom, uint32_t* __restrict__ rlen, uint32_t NRB) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, per = (NRB + 1023u) / 1024u;
    const uint32_t b = tid * per, e = min(b + per, NRB);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += rlen[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t i = b; i < e; ++i) { const uint32_t v = rlen[i]; rlen[i] = run; run += v; }
    if (tid == 1023u) geom[0] = part[1023];
}

// ---- the GEMM ---------------------------------------------------------------------------------------------------------------
struct ImgArgs {
    pbl_layer L;
    const _Float16* x;      // [M, K]
    void* y;                // [M, N] fp16 / fp32 / bf16
    int M, y_f32;
    const uint8_t* slots;   // the image's slots, its record starts, slot tables and level rows
    const uint32_t* rbase;
    const uint32_t* rtab;
    const uint32_t* levels;
    const float* tok_scale; // OM == 2: [M] fp32, the power of two (or +inf) every token's row of y is multiplied with (pbl_act_bf16_prepare)
    // XF (round 6): x as FRAGMENT-MAJOR copy (pbl_x_to_fragments): [token block of 32][8-column group][32 tokens][8 halves], rows of
    // xf_kp columns (K rounded up to 64, + 64 columns of zeros), token blocks up to a whole 256-token tile.  The MFMA waves then load
    // their B fragments straight from it (1 KiB contiguous per fragment) and no x tile goes through LDS at all.
    const _Float16* xf;
    uint32_t xf_kp;
    // the work of THIS launch (round 5, pbl_gemm_f16_image_ws: a thin last round is cut off and split along K): row tiles
    // [rt0, rt0 + nrt) x token tiles [tt0, tt0 + ntt), each in KSn work items of hps half slabs out of [h0, h0 + nh).  Work item
    // (tile, ks) writes y element (tok, row) to  ybase[(tok - ytok0) * ldy + (row - ycol0)],  ybase = y + ks * part_stride floats
    // (KSn > 1: fp32 partial tiles of a region buffer, summed by img_reduce_kernel; else the caller's y with ytok0 = ycol0 = 0, ldy = N).
    uint32_t rt0, nrt, tt0, ntt;
    int h0, nh, hps, KSn;
    size_t part_stride;
    uint32_t ldy, ycol0;
    int ytok0;
#if PBL_TRACE
    uint64_t* trace;
#endif
};

struct Frag { v8h a[4], b[2]; };

// fp32 -> bf16 bits, round to nearest even; NaN stays NaN (csrc/pbl_act.hip)
__device__ __forceinline__ uint32_t bf16_bits(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x0040u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// OM: the result's type -- 0 fp16, 1 fp32 (both exactly round 4's kernel), 2 bf16 with the per-token scale of bf16 activations:
// y[t, r] = bf16(acc * tok_scale[t] + bias[r]) (round 5: bf16 activations of the reference's perplexity / QAT loops,
// qat/run_qat.py:120, no longer leave the hand-written kernel for an unpack + library GEMM)
// XF (round 6): the x tile does not go through LDS.  Taking the loop apart (profiles/r06_gemm.md) showed the producer side alone at
// 53 us per round, 19 of them the LDS-DMA staging of x (2 MB per CU and tile at ~50 B/clk into the LDS, whatever the source) -- more
// than the dense library's whole kernel leaves.  With a fragment-major copy of x (one small kernel per DISTINCT x: q / k / v and
// gate / up share theirs) a B fragment is 1 KiB of contiguous memory: the MFMA waves load it into registers four k-steps ahead
// (asm loads, ONE counted `vmcnt(6)`: three k-steps of two loads stay in flight), the expanding waves only expand, and the LDS
// carries A alone.
template <int OM, bool KT, bool XF>
__global__ __launch_bounds__((GI_NCONS + GI_NPROD) * GW) void pbl_gemm_img_kernel(ImgArgs a) {
    constexpr bool Y32 = OM == 1;
    extern __shared__ __attribute__((aligned(16))) char smem_i[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const pbl_layer& L = a.L;
    const int K = int(L.K), M = a.M;
    const int NH = (K + GI_HS - 1) / GI_HS, NUt = (K + GI_XC - 1) / GI_XC;     // the layer's half slabs / 64-column steps
    // XCD-aware work order (speed only): workgroup b runs on XCD b % 8; every XCD gets a CONTIGUOUS range of the token-tile-major
    // work list, so the workgroups resident on an XCD share one 256-token slab of x in its L2
    const uint32_t nwg = gridDim.x;
    const uint32_t xq = nwg >> 3, xr_ = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const uint32_t wg = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + xi;
    // work item -> (K split, tile): splits outermost, then token tiles, then row tiles
    const uint32_t ntile = a.nrt * a.ntt;
    const int ks = int(wg / ntile);
    const uint32_t tl = wg - uint32_t(ks) * ntile;
    const uint32_t rowblk = a.rt0 + tl % a.nrt;
    const int tok0 = int(a.tt0 + tl / a.nrt) * GI_TOK;
    const int hb = a.h0 + ks * a.hps;                                  // this item's half slabs [hb, he)
    const int he = min(hb + a.hps, a.h0 + a.nh);
    const int ub = 2 * hb;                                             // ... = the layer's steps [ub, ub + NU)
    const int NU = min(NUt, 2 * he) - ub;
    char* const ybase = static_cast<char*>(a.y) + size_t(ks) * a.part_stride * sizeof(float);
    // ---- the result tile: Ys[256 tokens][128 rows] over the (then idle) ring and stages; the MFMA waves fill it, ALL eight waves
    // store it (32 tokens each, contiguous 16-byte units): with four storing waves the tail of the kernel was 6.4 us, with eight 3.4
    typedef typename std::conditional<Y32, float, _Float16>::type yt;
    constexpr uint32_t YSTR = Y32 ? 528u : 272u;              // bytes per token row: 128 rows + 16 B (every 16-byte read-back stays aligned)
    constexpr uint32_t YBIAS = GI_TOK * YSTR;                 // 128 floats of bias behind the tile (fp32 tile: 135168 + 512 <= 163840)
    static_assert(YBIAS + 512 <= GI_LDS, "tile + bias fit the workgroup's LDS");
    const uint32_t row0 = rowblk * GI_ROWS;
    auto store_tile = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // the tile is complete
        asm volatile("" ::: "memory");
        const bool vec = (a.ldy & (Y32 ? 3 : 7)) == 0 && row0 + GI_ROWS <= L.N;   // whole 16-byte units, all rows exist (row0 - ycol0: a multiple of 128)
        constexpr int UPR = GI_ROWS * int(sizeof(yt)) / 16;      // 16-byte units per token row: 16 (fp16) / 32 (fp32)
        constexpr int EPU = 16 / int(sizeof(yt));                // elements per unit
        for (int idx = lane; idx < 32 * UPR; idx += GW) {
            const int t = 32 * wave + idx / UPR, un = idx % UPR, tok = tok0 + t;
            if (tok >= M) continue;
            yt* dstg = reinterpret_cast<yt*>(ybase) + size_t(tok - a.ytok0) * a.ldy + (row0 - a.ycol0) + un * EPU;
            const yt* src = reinterpret_cast<const yt*>(smem_i + uint32_t(t) * YSTR) + un * EPU;
            if (vec) *reinterpret_cast<u32x4*>(dstg) = *reinterpret_cast<const u32x4*>(src);
            else
                for (int e_ = 0; e_ < EPU; ++e_)
                    if (row0 + un * EPU + e_ < L.N) dstg[e_] = src[e_];
        }
    };
#if PBL_TRACE
    uint64_t* tr = a.trace ? a.trace + (size_t(blockIdx.x) * 16 + wave) * 16 : nullptr;
    uint64_t tr_bar = 0, tr_vm = 0;
    TR_STAMP(0);
    if (tr && lane == 0) { tr[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tr[11] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
#endif

    if (wave >= GI_NCONS) {
        // =================================== expanding + staging waves ==============================================
        const int p = wave - GI_NCONS;
#if PBL_GEMM_PPRIO
        // (XF: measured flat -- 62.6 - 63.6 us for the expanding waves at priority 0 / 1 / 3 and the MFMA waves at 0 / 1 / 3, calls r6r, r6s)
        __builtin_amdgcn_s_setprio(PBL_GEMM_PPRIO);
#endif
        const uint32_t* levels = a.levels;
        const uint32_t gs = L.K / L.G;                          // columns per group (a multiple of 128)
        uint32_t rbv[2];
        const uint8_t* sbase[2];                                // the record's slots (wave uniform)
        uint32_t tabv[2][2];                                    // its slot table: lane l holds the words of half slabs l and 64 + l
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rbv[i] = min(rowblk * 8 + 2 * uint32_t(p) + uint32_t(i), L.NRB - 1);      // (a record beyond the layer mirrors the last one; its rows are never stored)
            sbase[i] = a.slots + size_t(__builtin_amdgcn_readfirstlane(a.rbase[rbv[i]])) * 256;
            tabv[i][0] = a.rtab[size_t(rbv[i]) * GI_TABW + lane];
            tabv[i][1] = a.rtab[size_t(rbv[i]) * GI_TABW + 64 + lane];
        }
        auto slot_word = [&](int i, int hh) -> uint32_t {         // rtab[record i][hh], hh uniform
            const uint32_t lo = __builtin_amdgcn_readlane(tabv[i][0], hh & 63), hi = __builtin_amdgcn_readlane(tabv[i][1], hh & 63);
            return hh < 64 ? lo : hi;
        };
        const uint32_t lane16 = uint32_t(lane) * 16u;
        uint32_t hl[2][16];                                     // (hi - lo : lo) of the 16 rows, current column group: SGPRs
        auto load_levels = [&](int i, uint32_t g) {
            const uint32_t* lp = levels + (size_t(rbv[i]) * L.G + g) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) hl[i][r] = __builtin_amdgcn_readfirstlane(lp[r]);
        };
        // slot registers: set s = (half slab & 1), record i, up to five 16-byte vectors.  A set is re-requested (for the half slab
        // after next) right after its last use and lands two half slabs later; nothing touches it in between (asm loads, counted
        // waits: see wait_all).  nvs: how many vectors the slot a set holds has (wave uniform).
        u32x4 e[2][2][GI_NVMAX];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int v = 0; v < GI_NVMAX; ++v) e[s_][i][v] = u32x4{0, 0, 0, 0};
        int nvs[2][2];
        // ONE asm block names all three vectors as read-write operands and skips the loads a smaller slot does not have with a
        // scalar branch INSIDE the block: with the loads under C++ branches the compiler merges the paths with register copies
        // (v_mov of registers whose load is in flight -- found by tools/audit_asm_loads.py in the first per-column build).
        auto request = [&](int i, int hh, u32x4 (&dst)[GI_NVMAX], int& nv_out) {
            const uint32_t t = slot_word(i, min(hh, NH - 1));
            const uint32_t nv = t >> 16;
            const uint8_t* sp = sbase[i] + size_t(t & 0xFFFFu) * 256 + 2048;      // (biased: the instruction offset is 13 bits, signed)
            const uint32_t lo = lane16;                         // vector v of the slot: one contiguous KiB, 16 bytes per lane
            static_assert(GI_NVMAX == 5, "the request block spells five loads out");
            asm volatile("global_load_dwordx4 %0, %5, %6 offset:-2048\n\t"
                         "s_cmp_lt_u32 %7, 2\n\t"
                         "s_cbranch_scc1 1f\n\t"
                         "global_load_dwordx4 %1, %5, %6 offset:-1024\n\t"
                         "s_cmp_lt_u32 %7, 3\n\t"
                         "s_cbranch_scc1 1f\n\t"
                         "global_load_dwordx4 %2, %5, %6\n\t"
                         "s_cmp_lt_u32 %7, 4\n\t"
                         "s_cbranch_scc1 1f\n\t"
                         "global_load_dwordx4 %3, %5, %6 offset:1024\n\t"
                         "s_cmp_lt_u32 %7, 5\n\t"
                         "s_cbranch_scc1 1f\n\t"
                         "global_load_dwordx4 %4, %5, %6 offset:2048\n"
                         "1:"
                         : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]) : "v"(lo), "s"(sp), "s"(nv) : "memory", "scc");
            nv_out = int(nv);
        };
        // XF: ALWAYS five loads per request -- without the eight x pieces per step between two requests the in-order count that proves
        // "the set requested four requests ago has landed" must come from the requests themselves: three requests x five loads =
        // `vmcnt(15)`.  A vector the slot does not have is loaded by LANE 0 ONLY (exec = 1 for that instruction, wave-uniform select, no
        // branch; the address falls back to the slot's first vector): it counts like any load and costs the memory pipe one 16-byte
        // access instead of a KiB -- with whole-wave re-reads of the first vector the kernel measured 2 us slower per 4096 x 4096 x 2048
        // call (the vector-memory path into registers is what bounds this kernel; call r6n).  Nobody stores those registers.
        auto request_xf = [&](int i, int hh, u32x4 (&dst)[GI_NVMAX], int& nv_out) {
            const uint32_t t = slot_word(i, min(hh, NH - 1));
            const uint32_t nv = t >> 16;
            const uint8_t* sp = sbase[i] + size_t(t & 0xFFFFu) * 256;
            const uint32_t o1 = lane16 + (nv > 1u ? 1024u : 0u), o2 = lane16 + (nv > 2u ? 2048u : 0u), o3 = lane16 + (nv > 3u ? 3072u : 0u);
            const uint32_t o4 = lane16 + (nv > 4u ? 4096u : 0u);
            uint64_t ex;
            static_assert(GI_NVMAX == 5, "the request block spells five loads out");
#if PBL_XF_REQ_PROBE          /* timing probe only (wrong results for slots with more vectors): two loads per request */
            asm volatile("global_load_dwordx4 %0, %5, %10\n\t"
                         "global_load_dwordx4 %1, %6, %10"
                         : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4])
                         : "v"(lane16), "v"(o1), "v"(o2), "v"(o3), "v"(o4), "s"(sp) : "memory");
            (void)ex;
#else
            asm volatile("global_load_dwordx4 %0, %6, %11\n\t"
                         "s_mov_b64 %5, exec\n\t"
                         "s_cmp_gt_u32 %12, 1\n\t"
                         "s_cselect_b64 exec, %5, 1\n\t"
                         "global_load_dwordx4 %1, %7, %11\n\t"
                         "s_cmp_gt_u32 %12, 2\n\t"
                         "s_cselect_b64 exec, %5, 1\n\t"
                         "global_load_dwordx4 %2, %8, %11\n\t"
                         "s_cmp_gt_u32 %12, 3\n\t"
                         "s_cselect_b64 exec, %5, 1\n\t"
                         "global_load_dwordx4 %3, %9, %11\n\t"
                         "s_cmp_gt_u32 %12, 4\n\t"
                         "s_cselect_b64 exec, %5, 1\n\t"
                         "global_load_dwordx4 %4, %10, %11\n\t"
                         "s_mov_b64 exec, %5"
                         : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "=&s"(ex)
                         : "v"(lane16), "v"(o1), "v"(o2), "v"(o3), "v"(o4), "s"(sp), "s"(nv) : "memory", "scc");
#endif
            nv_out = int(nv);
        };
        auto req = [&](int i, int hh, u32x4 (&dst)[GI_NVMAX], int& nv_out) {
            if constexpr (XF) request_xf(i, hh, dst, nv_out);
            else request(i, hh, dst, nv_out);
        };
        // x through a buffer descriptor that starts at this workgroup's first token: tokens >= M read zeros
        const char* xbase = reinterpret_cast<const char*>(a.x) + size_t(tok0) * size_t(K) * 2;
        const size_t xrem = size_t(min(M - tok0, GI_TOK)) * size_t(K) * 2;          // <= 256 * 32767 * 2 < 2^24
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase), 0, int(xrem), 0x00020000);
        // DMA piece q (1 KiB) of this wave: tokens 64 p + 8 q .. + 7, 128 B each; lane l lands on unit l & 7 of token (l >> 3),
        // which holds the LOGICAL unit (l & 7) ^ ((token >> 1) & 7)
        uint32_t xvoff[8], xvlast[KT ? 8 : 1];
        const uint32_t ktail_units = uint32_t(K & (GI_XC - 1)) >> 3;               // valid 16-byte units of the last step (0: no tail)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t tl = uint32_t(64 * p + 8 * q + (lane >> 3));
            const uint32_t lu = uint32_t(lane & 7) ^ ((tl >> 1) & 7);
            xvoff[q] = tl * uint32_t(K) * 2u + (lu << 4);
            // K % 64 != 0: the units of the LAST step beyond K would hold the next token row; their source offset is pushed out of
            // the descriptor's range instead, so they read zeros (the weights there are finite, the products vanish)
            if constexpr (KT) xvlast[q] = xvoff[q] + ((ktail_units && lu >= ktail_units) ? 0x40000000u : 0u);
        }
        auto stage_x = [&](int u, uint32_t slot_off) {           // the wave's 8 pieces of this item's step u into the ring slot at slot_off
            const uint32_t so = uint32_t(ub + u) * (GI_XC * 2);    // (steps past the item's end read the next split's columns: never used)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t dst = GI_X_OFF + slot_off + uint32_t(8 * p + q) * 1024u;
                // (analysis builds: bit 3 -- pieces 4 .. 7 re-read the bytes of pieces 0 .. 3, bit 4 -- every piece re-reads piece 0's: the
                // same instructions and LDS writes with half / an eighth of the distinct bytes from L2)
                uint32_t vo = xvoff[(PBL_IMG_ABLATE & 16) ? 0 : ((PBL_IMG_ABLATE & 8) ? (q & 3) : q)];
                if constexpr (KT) vo = (ub + u == NUt - 1) ? xvlast[q] : vo;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(smem_i + dst), 16, int(vo), int(so), 0, 0);
            }
        };
        // store addresses of the sign plane: lane l holds columns 2l, 2l+1 of all 16 rows in ONE dword (include/pbl.h: bit 16 e + pos);
        // row r's dword goes to  record + 256 r + (((l >> 2) ^ r) << 4) + 4 (l & 3)
        const uint32_t v0 = uint32_t(p) * 8192u + (uint32_t(lane >> 2) << 4) + (uint32_t(lane & 3) << 2);
        const uint32_t pb = uint32_t(p) * 8192u;
        auto expand = [&](int i, const u32x4 (&s)[GI_NVMAX], int nv, uint32_t stage) {
            const uint32_t base = stage * GI_AS_STAGE + uint32_t(i) * 4096u;       // (compile-time where the caller unrolls)
            const uint32_t d = s[0][0];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pos = r < 8 ? r + 8 : r - 8;
                const uint32_t m = (d >> pos) & 0x00010001u;
                uint32_t val;
                // per half: lo + bit * (hi - lo)  (mod 2^16): src1 = the pair's high half, src2 = its low half, for both lanes
                asm("v_pk_mad_u16 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(val) : "v"(m), "s"(hl[i][r]));
                *reinterpret_cast<uint32_t*>(smem_i + base + (v0 ^ uint32_t(0x110 * r))) = val;
            }
            asm volatile("" ::: "memory");                       // (the overlay comes after the plane: LDS keeps a wave's accesses in order)
            auto put_word = [&](uint32_t w) { *reinterpret_cast<uint16_t*>(smem_i + base + pb + (w >> 16)) = uint16_t(w & 0xFFFFu); };
            put_word(s[0][1]); put_word(s[0][2]); put_word(s[0][3]);
            if (nv >= 2) { put_word(s[1][0]); put_word(s[1][1]); put_word(s[1][2]); put_word(s[1][3]); }
            if (nv >= 3) { put_word(s[2][0]); put_word(s[2][1]); put_word(s[2][2]); put_word(s[2][3]); }
            if (nv >= 4) { put_word(s[3][0]); put_word(s[3][1]); put_word(s[3][2]); put_word(s[3][3]); }
            if (nv >= 5) { put_word(s[4][0]); put_word(s[4][1]); put_word(s[4][2]); put_word(s[4][3]); }
            asm volatile("" ::: "memory");
        };
        // `s_waitcnt vmcnt(10)` at the end of a step.  Issue order per step: 8 x pieces (D), then the slot request (R, 1 - 5 loads).
        // The barrier that follows publishes the x pieces of the PREVIOUS step, D(q-1); behind them came R(q-1) (>= 1), D(q) (8),
        // R(q) (>= 1): with at most 10 operations in flight D(q-1) has landed whatever the slot sizes are (in-order return) -- and
        // with it every slot set requested two or more steps ago.  (13 = 8 + the largest request looked right and let up to three
        // pieces of D(q-1) stay in flight when the slots are small: a race the config-3 test found on 2 of 256 workgroups.)  For
        // larger slots the constant is stricter than necessary (it then also covers part of R(q-1) / D(q)): those were issued a
        // whole step earlier.  ONE constant wait: a count that follows the slot sizes needs one asm statement per value under C++
        // branches, and the compiler merges those paths with register copies of slots still in flight
        // (tools/audit_asm_loads.py).  The slot registers are named as read-write operands, so no use can move above the wait and
        // no request below it.
#define GI_ESET(s_, i_) "+v"(e[s_][i_][0]), "+v"(e[s_][i_][1]), "+v"(e[s_][i_][2]), "+v"(e[s_][i_][3]), "+v"(e[s_][i_][4])
#define GI_EREGS GI_ESET(0, 0), GI_ESET(0, 1), GI_ESET(1, 0), GI_ESET(1, 1)
        auto wait_all = [&]() {
            TR_T0();
#if PBL_XF_REQ_PROBE
            if constexpr (XF) asm volatile("s_waitcnt vmcnt(6)" : GI_EREGS :: "memory");
#else
            if constexpr (XF) asm volatile("s_waitcnt vmcnt(15)" : GI_EREGS :: "memory");
#endif
            else asm volatile("s_waitcnt vmcnt(10)" : GI_EREGS :: "memory");
            TR_ADD(tr_vm);
        };
        auto barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's LDS stores are done
            TR_T0();
            __builtin_amdgcn_s_barrier();
            TR_ADD(tr_bar);
            asm volatile("" ::: "memory");
        };

        // ---- prologue: half slab 0 of A, steps 0 and 1 of x, the requests that follow
#pragma unroll
        for (int i = 0; i < 2; ++i) load_levels(i, (uint32_t(hb) * GI_HS) / gs);
#pragma unroll
        for (int i = 0; i < 2; ++i) req(i, hb, e[0][i], nvs[0][i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) req(i, hb + 1, e[1][i], nvs[1][i]);
        if constexpr (!XF) {
            stage_x(0, 0);
            stage_x(1, GI_XSLOT);
        }
        asm volatile("s_waitcnt vmcnt(0)" : GI_EREGS :: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) { expand(i, e[0][i], nvs[0][i], 0); req(i, hb + 2, e[0][i], nvs[0][i]); }
        TR_STAMP(1);
        barrier();                                               // barrier 0: stage 0 of A, steps 0 and 1 of x are in LDS

        // ---- step q (between barriers q and q + 1): x of step q + 2 into the slot step q - 1 used; record q & 1 of half slab
        // (q >> 1) + 1 into the stage half slab (q >> 1) - 1 used; that record's slot of half slab (q >> 1) + 3 requested
        uint32_t xs_free = 2 * GI_XSLOT, xs_a = 0, xs_b = GI_XSLOT;                 // ring slots: free now, then the two in use
        auto step = [&](int q, auto qm_tag) {
            constexpr int QM = decltype(qm_tag)::value;           // q & 3
            constexpr int i = QM & 1, st = ((QM >> 1) + 1) & 1;   // record; stage == register set == parity of the target half slab
            const int hh = hb + (q >> 1) + 1;                   // (absolute half slab; q counts this item's steps)
            if constexpr (!XF) {
                if (!(PBL_IMG_ABLATE & 2)) stage_x(q + 2, xs_free);
                { const uint32_t t = xs_free; xs_free = xs_a; xs_a = xs_b; xs_b = t; }
            }
            if (L.G > 1 && (uint32_t(hh) * GI_HS) % gs == 0 && hh < NH) load_levels(i, (uint32_t(hh) * GI_HS) / gs);
            if (!(PBL_IMG_ABLATE & 1)) expand(i, e[st][i], nvs[st][i], uint32_t(st));
            req(i, hh + 2, e[st][i], nvs[st][i]);
            wait_all();
            // XF: nothing but the A stages is shared, and a stage changes hands once per HALF SLAB -- the barrier behind an even step
            // (record 0 of the next stage written, record 1 not yet) orders nothing and is left out on both sides (32 instead of 64
            // workgroup barriers per 4096 columns).  With x in the ring every step publishes a slot of it.
            if constexpr (!XF || (QM & 1)) barrier();
        };
        int q = 0;
        for (; q + 4 <= NU - 1; q += 4) {
            step(q, std::integral_constant<int, 0>{});
            step(q + 1, std::integral_constant<int, 1>{});
            step(q + 2, std::integral_constant<int, 2>{});
            step(q + 3, std::integral_constant<int, 3>{});
        }
        if (q < NU - 1) {                                          // (nested: the control-flow graph then only has feasible paths, which is what
            step(q, std::integral_constant<int, 0>{});               // tools/audit_asm_loads.py walks)
            if (q + 1 < NU - 1) {
                step(q + 1, std::integral_constant<int, 1>{});
                if (q + 2 < NU - 1) step(q + 2, std::integral_constant<int, 2>{});
            }
        }
        TR_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (requests and x pieces past the end: nothing may land after the ring is reused)
        barrier();                                               // the MFMA waves are done with the ring: the result tile goes there
#undef GI_EREGS
#undef GI_ESET
#if PBL_TRACE
        if (tr && lane == 0) { tr[8] = tr_bar; tr[9] = tr_vm; }
#endif
        store_tile();                                            // (second barrier inside: the tile is complete) this wave's 32 tokens
        return;
    }

    // =================================== MFMA waves ==================================================================
    const int c = wave;
    const int i32 = lane & 31, g = lane >> 5;
    // fragment addresses: A unit (2 ks8 + g) ^ (i32 & 15) of row i32 (+ 32 rt), ks8 = 4 (u & 1) + k: the odd step's units are the
    // even one's with bit 3 flipped (byte 128); x unit (2 k + g) ^ ((i32 >> 1) & 7) of token 64 c + i32 (+ 32 tt)
    uint32_t aq[4], bq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) aq[k] = uint32_t(i32) * 256u + ((uint32_t(2 * k + g) ^ uint32_t(i32 & 15)) << 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) bq[k] = GI_X_OFF + uint32_t(64 * c + i32) * 128u + ((uint32_t(2 * k + g) ^ uint32_t((i32 >> 1) & 7)) << 4);

    v16f acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e_ = 0; e_ < 16; ++e_) acc[rt][tt][e_] = 0.f;

    float bias_lo = 0.f, bias_hi = 0.f;                           // bias of rows row0 + lane, row0 + 64 + lane (used in the epilogue)
    if (L.bias) {
        if (row0 + uint32_t(lane) < L.N) bias_lo = L.bias[row0 + lane];
        if (row0 + 64u + uint32_t(lane) < L.N) bias_hi = L.bias[row0 + 64 + lane];
    }
    // Fragment reads are issued from inline asm and waited for with counted lgkmcnt: left to itself hipcc sinks every ds_read to
    // just in front of the MFMA that needs it and waits lgkmcnt(0) there (seen in the ISA of the first build of this kernel: the
    // whole LDS latency in front of every k-step).  Here the six reads of k-step kk + 1 go out BEFORE the eight MFMAs of k-step kk;
    // the wait in front of those MFMAs leaves exactly the six younger reads outstanding (LDS returns in order).
    auto load_frag = [&](Frag& f, uint32_t aaddr, uint32_t baddr, auto aoff_tag) {
        constexpr int AO = decltype(aoff_tag)::value;              // stage offset of A (immediate)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[0]) : "v"(aaddr), "n"(AO) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=&v"(f.b[0]) : "v"(baddr) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(f.b[1]) : "v"(baddr) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[1]) : "v"(aaddr), "n"(AO + 8192) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[2]) : "v"(aaddr), "n"(AO + 16384) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[3]) : "v"(aaddr), "n"(AO + 24576) : "memory");
    };
    auto wait_frag = [&](Frag& f, auto n_tag) {                    // f has landed; n younger reads stay in flight
        constexpr int NOUT = decltype(n_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]) : "n"(NOUT) : "memory");
    };
    auto mma = [&](const Frag& f) {
        if (!(PBL_IMG_ABLATE & 4)) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    acc[rt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[rt], f.b[tt], acc[rt][tt], 0, 0, 0);
        }
        // nothing moves across: without it the scheduler sinks a k-step's MFMAs below the NEXT wait and even below the workgroup
        // barrier (seen in the ISA), and the matrix pipe idles exactly while the wave is parked
        __builtin_amdgcn_sched_barrier(0);
    };
    auto barrier0 = [&]() {
        TR_T0();
        __builtin_amdgcn_s_barrier();                             // barrier 0
        TR_ADD(tr_bar);
        asm volatile("" ::: "memory");
        TR_STAMP(1);
    };
    if constexpr (!XF) barrier0();
    if constexpr (XF) {
        // ---- XF: B fragments from the fragment-major copy of x, straight into registers.  k-step j (16 columns) of token block tb:
        // 1 KiB at  xf + tb * xf_kp * 64 + j * 1024,  lane l its 16 bytes at + 16 l.  bx[k][tt]: the fragment of the CURRENT step's
        // k-step k; the load for the same k of the NEXT step is issued right behind the MFMAs that consumed it (four k-steps ahead),
        // so in front of k-step j + 1 exactly the loads of j + 2, j + 3, j + 4 are younger: ONE constant `vmcnt(6)`.  The MFMA waves
        // issue no other vector-memory instruction.  Past the item's end the loads read the next split's columns or the copy's 64
        // columns of zero padding; nothing is used, `vmcnt(0)` behind the loop keeps them out of the epilogue's registers.
        struct FragA { v8h a[4]; };
        constexpr int XD = PBL_XF_DEPTH;                             // k-steps a B fragment is requested ahead: 4 (one step) or 8 (two steps)
        static_assert(XD == 4 || XD == 8, "B fragments are requested one or two 64-column steps ahead");
        v8h bx[XD][2];
        const uint64_t tb0 = uint64_t(tok0 / 32 + 2 * c);
        const char* bbase = reinterpret_cast<const char*>(a.xf) + tb0 * uint64_t(a.xf_kp) * 64u + uint64_t(ub) * 4096u;   // this item's first step
        const uint32_t off0 = uint32_t(lane) * 16u, off1 = off0 + a.xf_kp * 64u;
        auto load_b_ = [](v8h (&d)[2], const char* base, uint32_t o0, uint32_t o1, auto k_tag) {     // (offsets as arguments: clang does not
            constexpr int KO = decltype(k_tag)::value * 1024;                                          // capture locals named only by asm operands)
            asm volatile("global_load_dwordx4 %0, %2, %4 offset:%5" PBL_XF_BMOD "\n\t"
                         "global_load_dwordx4 %1, %3, %4 offset:%5" PBL_XF_BMOD
                         : "=&v"(d[0]), "=&v"(d[1]) : "v"(o0), "v"(o1), "s"(base), "n"(KO) : "memory");
        };
        auto load_b = [&](v8h (&d)[2], const char* base, auto k_tag) { load_b_(d, base, off0, off1, k_tag); };
        auto wait_b = [&](v8h (&d)[2]) {                            // d has landed; the 2 (XD - 1) younger loads stay in flight
            if constexpr (XD == 4) asm volatile("s_waitcnt vmcnt(6)" : "+v"(d[0]), "+v"(d[1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(14)" : "+v"(d[0]), "+v"(d[1]) :: "memory");
        };
        auto load_a = [&](FragA& f, uint32_t aaddr, auto aoff_tag) {
            constexpr int AO = decltype(aoff_tag)::value;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[0]) : "v"(aaddr), "n"(AO) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[1]) : "v"(aaddr), "n"(AO + 8192) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[2]) : "v"(aaddr), "n"(AO + 16384) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.a[3]) : "v"(aaddr), "n"(AO + 24576) : "memory");
        };
        auto wait_a = [&](FragA& f, auto n_tag) {
            constexpr int NOUT = decltype(n_tag)::value;
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]) : "n"(NOUT) : "memory");
        };
        auto mma_x = [&](const FragA& f, const v8h (&b)[2]) {
            if (!(PBL_IMG_ABLATE & 4)) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
                        acc[rt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[rt], b[tt], acc[rt][tt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        load_b(bx[0], bbase, std::integral_constant<int, 0>{});
        load_b(bx[1], bbase, std::integral_constant<int, 1>{});
        load_b(bx[2], bbase, std::integral_constant<int, 2>{});
        load_b(bx[3], bbase, std::integral_constant<int, 3>{});
        if constexpr (XD == 8) {
            load_b(bx[XD - 4], bbase + 4096, std::integral_constant<int, 0>{});
            load_b(bx[XD - 3], bbase + 4096, std::integral_constant<int, 1>{});
            load_b(bx[XD - 2], bbase + 4096, std::integral_constant<int, 2>{});
            load_b(bx[XD - 1], bbase + 4096, std::integral_constant<int, 3>{});
        }
        const char* bnext = bbase + (XD / 4) * 4096;
        barrier0();                                                  // (behind the first B requests: their latency hides in the expanding waves' prologue)
        FragA g0, g1;
        load_a(g0, aq[0], std::integral_constant<int, 0>{});
        auto substep_x = [&](auto um_tag, bool last) {             // last: the item's last step (its barrier frees the stages for the result tile)
            constexpr int UM = decltype(um_tag)::value;               // u & 3
            constexpr int abuf = ((UM >> 1) & 1) * GI_AS_STAGE, nabuf = (((UM + 1) >> 1) & 1) * GI_AS_STAGE;
            constexpr uint32_t ahalf = uint32_t(UM & 1) * 128u, nahalf = uint32_t((UM + 1) & 1) * 128u;
            constexpr int B0 = XD == 8 ? 4 * (UM & 1) : 0;            // this step's fragment registers (two steps ahead: even / odd steps alternate)
            load_a(g1, aq[1] ^ ahalf, std::integral_constant<int, abuf>{});
            wait_a(g0, std::integral_constant<int, 4>{});
            wait_b(bx[B0 + 0]);
            mma_x(g0, bx[B0 + 0]);
            load_b(bx[B0 + 0], bnext, std::integral_constant<int, 0>{});
            load_a(g0, aq[2] ^ ahalf, std::integral_constant<int, abuf>{});
            wait_a(g1, std::integral_constant<int, 4>{});
            wait_b(bx[B0 + 1]);
            mma_x(g1, bx[B0 + 1]);
            load_b(bx[B0 + 1], bnext, std::integral_constant<int, 1>{});
            load_a(g1, aq[3] ^ ahalf, std::integral_constant<int, abuf>{});
            wait_a(g0, std::integral_constant<int, 4>{});
            wait_b(bx[B0 + 2]);
            mma_x(g0, bx[B0 + 2]);
            load_b(bx[B0 + 2], bnext, std::integral_constant<int, 2>{});
            wait_a(g1, std::integral_constant<int, 0>{});             // every LDS read of this step has returned
            if ((UM & 1) || last) {                                   // (an even step stays inside its stage: no barrier, see the expanding waves)
                TR_T0();
                __builtin_amdgcn_s_barrier();                         // after an odd step: the next stage of A is complete, this one may be rewritten
                TR_ADD(tr_bar);
                asm volatile("" ::: "memory");
            }
            load_a(g0, aq[0] ^ nahalf, std::integral_constant<int, nabuf>{});
            wait_b(bx[B0 + 3]);
            mma_x(g1, bx[B0 + 3]);
            load_b(bx[B0 + 3], bnext, std::integral_constant<int, 3>{});
            if (!(PBL_IMG_ABLATE & 8)) bnext += 4096;          // (analysis builds, bit 3: every B load re-reads the same step -- L1 hits)
        };
        int u = 0;
        for (; u + 4 <= NU; u += 4) {
            substep_x(std::integral_constant<int, 0>{}, false);
            substep_x(std::integral_constant<int, 1>{}, false);
            substep_x(std::integral_constant<int, 2>{}, false);
            substep_x(std::integral_constant<int, 3>{}, false);
        }
        if (u < NU) { substep_x(std::integral_constant<int, 0>{}, u + 1 == NU); ++u; }
        if (u < NU) { substep_x(std::integral_constant<int, 1>{}, false); ++u; }
        if (u < NU) { substep_x(std::integral_constant<int, 2>{}, true); ++u; }
        wait_a(g0, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bx[0][0]), "+v"(bx[0][1]), "+v"(bx[1][0]), "+v"(bx[1][1]), "+v"(bx[2][0]), "+v"(bx[2][1]), "+v"(bx[3][0]), "+v"(bx[3][1]) :: "memory");
        if constexpr (XD == 8)
            asm volatile("" : "+v"(bx[XD - 4][0]), "+v"(bx[XD - 4][1]), "+v"(bx[XD - 3][0]), "+v"(bx[XD - 3][1]), "+v"(bx[XD - 2][0]), "+v"(bx[XD - 2][1]), "+v"(bx[XD - 1][0]), "+v"(bx[XD - 1][1]) :: "memory");
    } else {
    Frag f0, f1;
    uint32_t xs0 = 0, xs1 = GI_XSLOT, xs2 = 2 * GI_XSLOT;         // ring slot of this step, the next, the one after
    load_frag(f0, aq[0], bq[0], std::integral_constant<int, 0>{});
    // One 64-column step = 4 k-steps of 16 columns.  The barrier that opens step u + 1 sits in front of the step's LAST eight MFMAs:
    // they cover the first fragment reads of step u + 1.  Every step has it -- the last one's is the barrier behind which the
    // result tile may overwrite the ring (its prefetch then reads LDS nobody needs any more) -- so a step is branch free.
    auto substep = [&](auto um_tag) {
        constexpr int UM = decltype(um_tag)::value;               // u & 3
        constexpr int abuf = ((UM >> 1) & 1) * GI_AS_STAGE, nabuf = (((UM + 1) >> 1) & 1) * GI_AS_STAGE;
        constexpr uint32_t ahalf = uint32_t(UM & 1) * 128u, nahalf = uint32_t((UM + 1) & 1) * 128u;
        load_frag(f1, aq[1] ^ ahalf, bq[1] + xs0, std::integral_constant<int, abuf>{});
        wait_frag(f0, std::integral_constant<int, 6>{});
        mma(f0);
        load_frag(f0, aq[2] ^ ahalf, bq[2] + xs0, std::integral_constant<int, abuf>{});
        wait_frag(f1, std::integral_constant<int, 6>{});
        mma(f1);
        load_frag(f1, aq[3] ^ ahalf, bq[3] + xs0, std::integral_constant<int, abuf>{});
        wait_frag(f0, std::integral_constant<int, 6>{});
        mma(f0);
        wait_frag(f1, std::integral_constant<int, 0>{});          // every read of this step has returned
        {
            TR_T0();
            __builtin_amdgcn_s_barrier();                         // barrier u + 1: step u + 1 of x (and, after an odd step, the next stage of A) is complete
            TR_ADD(tr_bar);
            asm volatile("" ::: "memory");
        }
        load_frag(f0, aq[0] ^ nahalf, bq[0] + xs1, std::integral_constant<int, nabuf>{});
        mma(f1);
        { const uint32_t t = xs0; xs0 = xs1; xs1 = xs2; xs2 = t; }
    };
    int u = 0;
    for (; u + 4 <= NU; u += 4) {
        substep(std::integral_constant<int, 0>{});
        substep(std::integral_constant<int, 1>{});
        substep(std::integral_constant<int, 2>{});
        substep(std::integral_constant<int, 3>{});
    }
    if (u < NU) { substep(std::integral_constant<int, 0>{}); ++u; }
    if (u < NU) { substep(std::integral_constant<int, 1>{}); ++u; }
    if (u < NU) { substep(std::integral_constant<int, 2>{}); ++u; }
    wait_frag(f0, std::integral_constant<int, 0>{});              // (the last prefetch: nothing may land in the registers later)
    }
    TR_STAMP(2);
#if PBL_TRACE
    if (tr && lane == 0) { tr[8] = tr_bar; tr[9] = tr_vm; }
#endif

    // ---- epilogue: accumulators (+ bias) -> the tile; the bias of the 128 rows goes through LDS (two coalesced loads per lane,
    // requested before the loop) instead of 64 predicated loads per lane
    if (L.bias) {
        reinterpret_cast<float*>(smem_i + YBIAS)[lane] = bias_lo;
        reinterpret_cast<float*>(smem_i + YBIAS)[64 + lane] = bias_hi;     // (every MFMA wave writes the same 128 values)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        float sc = 1.f;                                           // OM == 2: this lane's token of the block (tokens >= M are never stored)
        if constexpr (OM == 2) { const int tok = tok0 + 64 * c + 32 * tt + i32; sc = tok < M ? a.tok_scale[tok] : 0.f; }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int rloc = rt * 32 + 8 * q4 + 4 * g;        // 4 consecutive rows held by this lane: D row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                v4f b4 = {0.f, 0.f, 0.f, 0.f};
                if (L.bias) b4 = *reinterpret_cast<const v4f*>(smem_i + YBIAS + uint32_t(rloc) * 4u);
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = OM == 2 ? __builtin_fmaf(acc[rt][tt][4 * q4 + r], sc, b4[r]) : acc[rt][tt][4 * q4 + r] + b4[r];
                char* dst = smem_i + uint32_t(64 * c + 32 * tt + i32) * YSTR + uint32_t(rloc) * sizeof(yt);
                if (Y32) *reinterpret_cast<v4f*>(dst) = v4f{o[0], o[1], o[2], o[3]};
                else {
                    uint2 pk;
                    if constexpr (OM == 2) {
                        pk.x = bf16_bits(o[0]) | (bf16_bits(o[1]) << 16);
                        pk.y = bf16_bits(o[2]) | (bf16_bits(o[3]) << 16);
                    } else {
                        pk.x = h16(o[0]) | (h16(o[1]) << 16);
                        pk.y = h16(o[2]) | (h16(o[3]) << 16);
                    }
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            }
    }
    store_tile();
#if PBL_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_STAMP(3);
#endif
}

// ---- <= 64 rows of x over the same image ------------------------------------------------------------------------------------
// Replaces the same call at 5 - 64 rows (quant/outlier_quantizer.py:101-106 under a small serving batch; BASELINE.json configs[3]).
// HBM-bound: the image is read once, x (32 or 64 rows x K fp16) lives in L2.  No roles among the working waves: every WAVE owns a
// pair of records (32 rows) and the workgroup's range of half slabs, and per half slab
//   * rebuilds its two records' 16 x 128 fp16 tiles in its PRIVATE 8 KiB of LDS from the slots (plane dword -> 16 stores, every
//     entry word one 2-byte store: the GEMM kernel's expansion), which arrive through a ring of four slot register sets requested
//     two half slabs ahead (plain loads: the compiler counts vmcnt),
//   * reads the A fragments back (a wave's LDS operations execute in order: no wait between the stores and the reads) and
//     multiplies them with x's 32 x 128 tile (two of them for 33 - 64 rows), which a FIFTH wave of the workgroup stages by LDS-DMA
//     (eight 1 KiB pieces per tile, whole 128-byte lines of x; a fragment read straight from global memory touches 32 lines per
//     instruction) into a double buffer: ONE workgroup barrier per half slab.  The staging wave keeps the x pieces out of the
//     working waves' vmcnt queue (memory operations return in order: waiting for a piece also waits for every slot request issued
//     before it).
// K is split over gridDim.y; the splits' fp32 partial tiles are added in split order by sb_reduce_kernel (deterministic): a second
// small launch, 2.9 us behind the first.  (Folding the sum into the LAST split to arrive was built and measured, call r4-23/24: with
// release / acquire fences at agent scope -- a buffer_wbl2 per wave -- 44 - 110 us per launch instead of 25; with agent-scope
// stores and loads of the partial tiles and no fence 36 - 61 us: traffic past the XCD's L2 costs more than the launch.)
struct SbArgs {
    pbl_layer L;
    const _Float16* x;      // [M, K]
    void* y;                // [M, N] fp16 / fp32 (KS == 1)
    float* part;            // [KS][M][N] fp32 (KS > 1)
    int M, y_f32, KS, hps;  // hps: half slabs per K split
    const uint8_t* slots;   // the image's slots, its record starts, slot tables and level rows
    const uint32_t* rbase;
    const uint32_t* rtab;
    const uint32_t* levels;
};
#define SB_WAVES 4                   // working waves of a round-4 workgroup (+ 1 that stages x): RP = 4 row pairs, KQ = 1
#ifndef PBL_SB_WGS_PER_CU
#define PBL_SB_WGS_PER_CU 2          // workgroups per CU the K split of the round-4 geometry aims at
#endif
// Round 6 ("remove the K split instead of tuning it"): the workgroup's geometry is a template parameter pair --
// RP row pairs x KQ K PHASES.  Working wave w owns row pair w % RP and phase w / RP: of the workgroup's range of half slabs it takes
// h0 + KQ j + phase (j = 0, 1, ...: the phases interleave, so the workgroup as a whole streams each record's slots front to back and
// the staging wave's x tiles of a step are one contiguous range of columns).  The KQ phases' accumulators of a row pair are added
// through LDS in phase order at the end (deterministic), so a layer with enough rows needs NO split across workgroups: no partial
// tiles in HBM, no workspace, no second launch (13824 x 5120: 216 workgroups of 2 x 4 waves instead of 432 of 4 + a reduce launch
// over 4 x 1.77 MB of partials).  Layers with few rows keep the round-4 geometry and its split across gridDim.y.
#define SB_X_OFF(RP_, KQ_) ((RP_) * (KQ_) * 8192)
#define SB_LDS(NTB_, RP_, KQ_) (SB_X_OFF(RP_, KQ_) + 2 * (KQ_) * 8192 * (NTB_))   // + the x tiles' double buffer: KQ x (32 NTB tokens x 128 columns)

typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;        // constant address space: scalar loads
// Half slabs of slot requests in flight per wave (2 sets each).  Measured on 13824 x 5120, 20 % salients, 32 rows (calls r4-30 .. 32):
// depth 2 at <= 128 VGPRs (four waves per SIMD: two 5-wave workgroups always find room on a CU) 21.9 us; depth 2 at <= 168 VGPRs
// 28.7 us and depth 4 (164 VGPRs) 33.1 us -- three waves per SIMD are 12 slots, but a second 5-wave workgroup only fits when its
// waves fall on the right SIMDs, and with one workgroup per CU the waves' own instruction streams (not the bytes in flight) bound
// the kernel.  So: depth 2, and the register budget of four waves per SIMD wherever the sets allow it.
#ifndef PBL_SB_DEPTH
#define PBL_SB_DEPTH 2
#endif
#ifndef PBL_SB_WPE
#define PBL_SB_WPE(NVK_, NTB_) ((NTB_) == 1 && (NVK_) <= 3 ? 4 : 3)   // (what the register allocator reaches: 33 - 64 rows hold two accumulator blocks)
#endif
#ifndef PBL_SB_WPE_KQ
// the K-phase geometries are limited by their LDS, not their registers: 2 x 4 is ONE workgroup of 9 waves per CU (three waves on one
// SIMD: <= 168 VGPRs), 1 x 3 two workgroups of 4 waves (two per SIMD)
#define PBL_SB_WPE_KQ(RP_, KQ_) ((RP_) * (KQ_) + 1 > 8 ? 3 : 2)
#endif
#ifndef PBL_SB_NT
#define PBL_SB_NT 0                  // slot loads with the non-temporal hint (the image is read once)
#endif
// NTB: blocks of 32 rows of x (1: up to 32 rows, 2: up to 64 -- the image is still read once; twice the x tile, accumulators, MFMAs)
template <int NVK, bool KT, int NTB, int RP, int KQ>
__global__ __launch_bounds__((RP * KQ + 1) * GW) __attribute__((amdgpu_waves_per_eu(KQ > 1 ? PBL_SB_WPE_KQ(RP, KQ) : PBL_SB_WPE(NVK, NTB), KQ > 1 ? PBL_SB_WPE_KQ(RP, KQ) : PBL_SB_WPE(NVK, NTB))))
void pbl_sb_img_kernel(SbArgs a) {
    constexpr int D = PBL_SB_DEPTH;
    static_assert(D >= 2 && D % 2 == 0, "the x double buffer's parity is static in the unrolled loop");
    constexpr int NWORK = RP * KQ;
    __shared__ __attribute__((aligned(16))) char smem_s[SB_LDS(NTB, RP, KQ)];
    constexpr uint32_t XBUF = 8192u * NTB;                    // one x tile
    constexpr uint32_t XOFF = SB_X_OFF(RP, KQ);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const pbl_layer& L = a.L;
    const int K = int(L.K), M = a.M;
    const int NH = (K + GI_HS - 1) / GI_HS;
    const int ks = int(blockIdx.y);
    const int h0 = ks * a.hps, h1 = min(h0 + a.hps, NH);
    const int nsteps = (h1 - h0 + KQ - 1) / KQ;               // steps of the workgroup: every phase takes (at most) one half slab per step
    const int nstepsp = (nsteps + D - 1) / D * D;             // (the working waves run whole rounds of their slot ring: one barrier per step of those)
    if (wave == NWORK) {
        // ---- the staging wave.  x tile of a half slab in LDS: [32 tokens][16 units of 8 columns], unit u of token t at
        // 256 t + 16 (u ^ (t & 15)).  A DMA piece is 1 KiB = 4 token rows: lane l lands on unit (l & 15) of token 4 piece + (l >> 4),
        // which holds the LOGICAL unit (l & 15) ^ (token & 15).  Through a buffer descriptor over the M rows of x: tokens >= M read zeros.
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, int(size_t(M) * size_t(K) * 2), 0x00020000);
        uint32_t xvoff[8 * NTB], xvlast[KT ? 8 * NTB : 1];
        const uint32_t ktail_units = uint32_t(K & (GI_HS - 1)) >> 3;               // valid units of the last half slab (0: no tail)
#pragma unroll
        for (int q = 0; q < 8 * NTB; ++q) {
            const uint32_t tl = uint32_t(4 * q + (lane >> 4));
            const uint32_t lu = uint32_t(lane & 15) ^ (tl & 15);
            xvoff[q] = tl * uint32_t(K) * 2u + (lu << 4);
            // the units of the LAST half slab beyond K would hold the next token row: pushed out of the descriptor's range, they read zeros
            if constexpr (KT) xvlast[q] = xvoff[q] + ((ktail_units && lu >= ktail_units) ? 0x40000000u : 0u);
        }
        for (int st = 0; st <= nstepsp; ++st) {              // x of step st (KQ half slabs) into buffer st & 1, then the barrier the others open step st behind
            if (st < nsteps) {
                const uint32_t buf = uint32_t(st) & 1u;
#pragma unroll
                for (int ph = 0; ph < KQ; ++ph) {
                    const int h = h0 + st * KQ + ph;
                    if (h >= h1) break;                      // (the last step of a range that is not a multiple of KQ: those phases idle)
#pragma unroll
                    for (int q = 0; q < 8 * NTB; ++q) {
                        uint32_t vo = xvoff[q];
                        if constexpr (KT) vo = (h == NH - 1) ? xvlast[q] : vo;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(smem_s + XOFF + (buf * KQ + uint32_t(ph)) * XBUF + uint32_t(q) * 1024u), 16, int(vo),
                                                                 h * (GI_HS * 2), 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // barrier st: x(st) is complete; everyone has left the buffer x(st + 1) goes to
            asm volatile("" ::: "memory");
        }
        if constexpr (KQ > 1) {                              // the two barriers of the phases' reduction (every wave of the workgroup takes them)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    const int rp = KQ > 1 ? wave % RP : wave, kq = KQ > 1 ? wave / RP : 0;      // this wave's row pair of the workgroup, its K phase
    const uint32_t npairs = (L.NRB + 1) / 2;
    const uint32_t pair_raw = blockIdx.x * RP + uint32_t(rp);
    const uint32_t pair = min(pair_raw, npairs - 1);         // (a surplus wave mirrors the last pair: it keeps the barriers and stores nothing)
    char* const As = smem_s + wave * 8192;
    const uint32_t* levels = a.levels;
    const uint32_t gs = L.K / L.G;
    uint32_t rbv[2];
    const uint8_t* sbase[2];
    uint32_t tabv[2][2];                                     // the records' slot tables: lane l holds the words of half slabs l and 64 + l
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rbv[i] = min(2 * pair + uint32_t(i), L.NRB - 1);     // (an odd record count: the last pair mirrors the last record, whose rows are stored once)
        sbase[i] = a.slots + size_t(__builtin_amdgcn_readfirstlane(a.rbase[rbv[i]])) * 256;
        tabv[i][0] = a.rtab[size_t(rbv[i]) * GI_TABW + lane];
        tabv[i][1] = a.rtab[size_t(rbv[i]) * GI_TABW + 64 + lane];
    }
    auto slot_word = [&](int i, int hh) -> uint32_t {          // rtab[record i][hh], hh uniform
        const uint32_t lo = __builtin_amdgcn_readlane(tabv[i][0], hh & 63), hi = __builtin_amdgcn_readlane(tabv[i][1], hh & 63);
        return hh < 64 ? lo : hi;
    };
    uint32_t hl[2][16];
    auto load_levels = [&](int i, uint32_t g) {
        const_u32_ptr lp = (const_u32_ptr)(reinterpret_cast<uintptr_t>(levels + (size_t(rbv[i]) * L.G + g) * 16));   // (the image is read-only here)
#pragma unroll
        for (int r = 0; r < 16; ++r) hl[i][r] = lp[r];
    };
    // ring of 2 D slot sets: record i of half slab h sits in set 2 ((h - h0) % D) + i, requested D half slabs ahead; nvs: the
    // vectors of the slot a set holds
    u32x4 e[2 * D][NVK];
    uint32_t nvs[2 * D];
    const uint32_t lane16 = uint32_t(lane) * 16u;
    auto request = [&](int i, int h, u32x4 (&dst)[NVK], uint32_t& nv_out) {
        const uint32_t t = slot_word(i, min(h, h1 - 1));     // (past the split's end: its last slot again -- a cache hit, never used)
        const uint32_t nv = t >> 16;
        const uint8_t* sp = sbase[i] + size_t(t & 0xFFFFu) * 256 + lane16;
#pragma unroll
        for (int v = 0; v < NVK; ++v)                           // (a smaller slot: its last vector again, a cache hit; not stored)
#if PBL_SB_NT
            dst[v] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sp + 1024u * min(uint32_t(v), nv - 1u)));
#else
            dst[v] = *reinterpret_cast<const u32x4*>(sp + 1024u * min(uint32_t(v), nv - 1u));
#endif
        nv_out = nv;
    };
    const uint32_t v0 = (uint32_t(lane >> 2) << 4) + (uint32_t(lane & 3) << 2);
    auto expand = [&](int i, const u32x4 (&s)[NVK], uint32_t nv) {
        char* base = As + i * 4096;
        const uint32_t d = s[0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pos = r < 8 ? r + 8 : r - 8;
            const uint32_t m = (d >> pos) & 0x00010001u;
            uint32_t val;
            asm("v_pk_mad_u16 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(val) : "v"(m), "s"(hl[i][r]));
            *reinterpret_cast<uint32_t*>(base + (v0 ^ uint32_t(0x110 * r))) = val;
        }
        asm volatile("" ::: "memory");                       // (the overlay comes after the plane: in order)
        auto put_word = [&](uint32_t w) { *reinterpret_cast<uint16_t*>(base + (w >> 16)) = uint16_t(w & 0xFFFFu); };
        put_word(s[0][1]); put_word(s[0][2]); put_word(s[0][3]);
#pragma unroll
        for (int v = 1; v < NVK; ++v)
            if (nv > uint32_t(v)) { put_word(s[v][0]); put_word(s[v][1]); put_word(s[v][2]); put_word(s[v][3]); }
        asm volatile("" ::: "memory");
    };
    const int i32 = lane & 31, g = lane >> 5;
    uint32_t aq[8], bq[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
        aq[k8] = uint32_t(i32) * 256u + ((uint32_t(2 * k8 + g) ^ uint32_t(i32 & 15)) << 4);
        bq[k8] = XOFF + uint32_t(kq) * XBUF + aq[k8];        // (the x tile has the A tile's geometry: 32 rows of 256 bytes, the same swizzle; this phase's tile of a buffer)
    }
    v16f acc[NTB];
#pragma unroll
    for (int b = 0; b < NTB; ++b)
#pragma unroll
        for (int e_ = 0; e_ < 16; ++e_) acc[b][e_] = 0.f;

    // this wave's half slabs: hs(j) = h0 + KQ j + kq, j = 0 .. nsteps - 1 (live while < h1; the last step of a range that is not a
    // multiple of KQ idles the higher phases)
    const int hfirst = min(h0 + kq, h1 - 1);
    uint32_t gcur = (uint32_t(hfirst) * GI_HS) / gs;          // the column group whose levels the SGPRs hold
    load_levels(0, gcur); load_levels(1, gcur);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        // (in THIS order: the compiler counts vmcnt for the loop's first use of a set from the loads issued behind it on every path
        // into the loop; left free, the scheduler puts set 0's loads last in the prologue and the loop then waits for all but five)
        request(0, h0 + KQ * j + kq, e[2 * j], nvs[2 * j]);
        asm volatile("" ::: "memory");
        request(1, h0 + KQ * j + kq, e[2 * j + 1], nvs[2 * j + 1]);
        asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                            // barrier 0: x of the first step is in LDS
    asm volatile("" ::: "memory");
    // `live`: false for the steps that pad the last round of the ring, and for a phase beyond the range's end (wave uniform).  They
    // keep the round's shape -- the same loads, the same barrier -- and skip the work: with the SAME number of loads on every path
    // through the loop the compiler's vmcnt counts stay at "all younger sets in flight" (an early exit from the round, or a
    // remainder behind the loop, made it size the round's first wait for the shortest path: vmcnt(5) instead of vmcnt(21)).
    // The fragment reads are issued from inline asm one k-step ahead of the MFMA that consumes them and waited for with a counted
    // lgkmcnt (LDS returns in order; the reads come after this wave's tile stores in program order, and a wave's LDS operations
    // execute in order): round 4 left them to the compiler, which kept the pipelined form (`lgkmcnt(2)`) in some builds and fell back
    // to read - wait(0) - multiply per k-step in others (round 6: any change to the loop around it) -- a full LDS latency in front of
    // each of the eight MFMAs of a half slab, on the critical path of a wave that has few neighbours to hide behind.
    struct SbFrag { v8h a, b[NTB]; };
    const uint32_t a_off = uint32_t(wave) * 8192u;
    auto half_slab = [&](int h, auto buf_tag, u32x4 (&sa)[NVK], u32x4 (&sb)[NVK], uint32_t& nva, uint32_t& nvb, bool live) {
        constexpr uint32_t XB0 = uint32_t(decltype(buf_tag)::value) * (KQ * XBUF);     // this step's x buffer (immediate offset)
        if (live && L.G > 1) {
            const uint32_t gg = (uint32_t(h) * GI_HS) / gs;
            if (gg != gcur) { gcur = gg; load_levels(0, gg); load_levels(1, gg); }
        }
        if (live) expand(0, sa, nva);
        request(0, h + KQ * D, sa, nva);
        if (live) expand(1, sb, nvb);
        request(1, h + KQ * D, sb, nvb);
        if (live) {
            SbFrag f[2];
            auto load_frag = [&](SbFrag& d, int k8) {
                asm volatile("ds_read_b128 %0, %1" : "=&v"(d.a) : "v"(a_off + aq[k8]) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d.b[0]) : "v"(bq[k8]), "n"(XB0) : "memory");
                if constexpr (NTB == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d.b[1]) : "v"(bq[k8]), "n"(XB0 + 8192u) : "memory");
            };
            auto wait_frag = [&](SbFrag& d, auto n_tag) {                  // d has landed; n younger reads stay in flight
                constexpr int NOUT = decltype(n_tag)::value;
                if constexpr (NTB == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(d.a), "+v"(d.b[0]), "+v"(d.b[1]) : "n"(NOUT) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(d.a), "+v"(d.b[0]) : "n"(NOUT) : "memory");
            };
            load_frag(f[0], 0);
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                if (k8 < 7) {
                    load_frag(f[(k8 + 1) & 1], k8 + 1);
                    wait_frag(f[k8 & 1], std::integral_constant<int, 1 + NTB>{});
                } else {
                    wait_frag(f[k8 & 1], std::integral_constant<int, 0>{});
                }
#pragma unroll
                for (int b = 0; b < NTB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k8 & 1].a, f[k8 & 1].b[b], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                         // (nothing moves across: the MFMAs stay between their wait and the next one)
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of the x tile are done; then everyone's, and the next tile is in
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // (the induction variable is the round's first half slab of PHASE 0 -- the same trip count, i.e. the same barriers, for every phase;
    // written over the step index instead, the compiler kept 24 more VGPRs live and the 128-register variants spilled)
    static_assert(D == 2 || D == 4, "the ring's steps are spelled out (their x buffer is an immediate offset)");
    for (int hb = h0; hb < h1; hb += KQ * D) {
        half_slab(hb + kq, std::integral_constant<int, 0>{}, e[0], e[1], nvs[0], nvs[1], hb + kq < h1);
        half_slab(hb + KQ + kq, std::integral_constant<int, 1>{}, e[2], e[3], nvs[2], nvs[3], hb + KQ + kq < h1);
        if constexpr (D == 4) {
            half_slab(hb + 2 * KQ + kq, std::integral_constant<int, 0>{}, e[4], e[5], nvs[4], nvs[5], hb + 2 * KQ + kq < h1);
            half_slab(hb + 3 * KQ + kq, std::integral_constant<int, 1>{}, e[6], e[7], nvs[6], nvs[7], hb + 3 * KQ + kq < h1);
        }
    }

    if constexpr (KQ > 1) {
        // ---- the phases of a row pair added in phase order (deterministic): phases 1 .. KQ - 1 park their accumulators in their own
        // (now idle) A tile, 4 KiB per token block, element e of lane l at 256 e + 4 l; phase 0 adds them and stores
        float* mine = reinterpret_cast<float*>(As);
        if (kq > 0) {
#pragma unroll
            for (int b = 0; b < NTB; ++b)
#pragma unroll
                for (int e_ = 0; e_ < 16; ++e_) mine[(b * 16 + e_) * 64 + lane] = acc[b][e_];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kq == 0) {
#pragma unroll
            for (int ph = 1; ph < KQ; ++ph) {
                const float* other = reinterpret_cast<const float*>(smem_s + (ph * RP + rp) * 8192);
#pragma unroll
                for (int b = 0; b < NTB; ++b)
#pragma unroll
                    for (int e_ = 0; e_ < 16; ++e_) acc[b][e_] += other[(b * 16 + e_) * 64 + lane];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (the second barrier the staging wave takes: every wave leaves together)
        asm volatile("" ::: "memory");
        if (kq > 0) return;
    }

    // ---- the 32 x 32 tiles: rows (reg & 3) + 8 (reg >> 2) + 4 g of the pair, token 32 b + i32
    const uint32_t row0 = pair_raw * 32u;
    if (pair_raw >= npairs) return;
    const bool whole = row0 + 32 <= L.N && (L.N & 3) == 0;
#pragma unroll
    for (int b = 0; b < NTB; ++b) {
        const int tok = 32 * b + i32;
        if (tok >= M) continue;
        float o[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t row = row0 + uint32_t(8 * (q >> 2) + 4 * g + (q & 3));
            o[q] = acc[b][q];
            if (L.bias && ks == 0 && row < L.N) o[q] += L.bias[row];
        }
        if (a.KS > 1) {                                       // this split's partial tile; sb_reduce_kernel adds the splits
            float* dst = a.part + (size_t(ks) * M + tok) * L.N + row0 + 4 * g;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                if (whole) *reinterpret_cast<v4f*>(dst + 8 * q4) = v4f{o[4 * q4], o[4 * q4 + 1], o[4 * q4 + 2], o[4 * q4 + 3]};
                else
                    for (int r = 0; r < 4; ++r) if (row0 + 4 * g + 8 * q4 + r < L.N) dst[8 * q4 + r] = o[4 * q4 + r];
            }
            continue;
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const uint32_t row = row0 + uint32_t(8 * q4 + 4 * g);
            if (a.y_f32) {
                float* dst = static_cast<float*>(a.y) + size_t(tok) * L.N + row;
                if (whole) *reinterpret_cast<v4f*>(dst) = v4f{o[4 * q4], o[4 * q4 + 1], o[4 * q4 + 2], o[4 * q4 + 3]};
                else
                    for (int r = 0; r < 4; ++r) if (row + r < L.N) dst[r] = o[4 * q4 + r];
            } else {
                _Float16* dst = static_cast<_Float16*>(a.y) + size_t(tok) * L.N + row;
                if (whole) {
                    uint2 pk;
                    pk.x = h16(o[4 * q4]) | (h16(o[4 * q4 + 1]) << 16); pk.y = h16(o[4 * q4 + 2]) | (h16(o[4 * q4 + 3]) << 16);
                    *reinterpret_cast<uint2*>(dst) = pk;
                } else
                    for (int r = 0; r < 4; ++r) if (row + r < L.N) dst[r] = _Float16(o[4 * q4 + r]);
            }
        }
    }
}

// y = the K splits' partial tiles added in split order; 4 elements per thread (MN % 4 == 0 whenever N % 4 == 0).  All splits'
// values are requested together (ONE memory latency: the kernel is nothing but latency), up to 16 at a time.
// ACT (bf16 activations, round 5): y = cast(sum * tok_scale[token] + bias[row]) -- pbl_act_finish folded into the reduce (out: 1 fp32,
// 2 bf16); the plain form (ACT == false) is round 4's kernel.
template <bool ACT>
__global__ __launch_bounds__(256) void sb_reduce_kernel(const float* __restrict__ part, void* __restrict__ y, int KS, size_t MN, int y_f32,
                                                        const float* __restrict__ tok_scale, const float* __restrict__ bias, uint32_t N) {
    const size_t i = (size_t(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    if (ACT ? !(N & 3) : !(MN & 3)) {
        v4f sum = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < KS; k0 += 16) {
            v4f v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = k0 + j < KS ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(part + size_t(k0 + j) * MN + i)) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; ++j) if (k0 + j < KS) sum = (k0 + j) ? sum + v[j] : v[j];
        }
        if constexpr (ACT) {                                  // (N % 4 == 0 here: the four elements are one token's)
            const size_t t = i / N;
            const uint32_t r = uint32_t(i - t * N);
            const float sc = tok_scale[t];
            v4f b = {0.f, 0.f, 0.f, 0.f};
            if (bias) b = v4f{bias[r], bias[r + 1], bias[r + 2], bias[r + 3]};
#pragma unroll
            for (int e_ = 0; e_ < 4; ++e_) sum[e_] = __builtin_fmaf(sum[e_], sc, b[e_]);
            if (y_f32 == 1) *reinterpret_cast<v4f*>(static_cast<float*>(y) + i) = sum;
            else {
                uint2 pk;
                pk.x = bf16_bits(sum[0]) | (bf16_bits(sum[1]) << 16); pk.y = bf16_bits(sum[2]) | (bf16_bits(sum[3]) << 16);
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(y) + i) = pk;
            }
            return;
        }
        if (y_f32) *reinterpret_cast<v4f*>(static_cast<float*>(y) + i) = sum;
        else {
            uint2 pk;
            pk.x = h16(sum[0]) | (h16(sum[1]) << 16); pk.y = h16(sum[2]) | (h16(sum[3]) << 16);
            *reinterpret_cast<uint2*>(static_cast<_Float16*>(y) + i) = pk;
        }
        return;
    }
    for (size_t j = i; j < MN && j < i + 4; ++j) {
        float sum = part[j];
        for (int k = 1; k < KS; ++k) sum += part[size_t(k) * MN + j];
        if constexpr (ACT) {
            const size_t t = j / N;
            sum = __builtin_fmaf(sum, tok_scale[t], bias ? bias[j - t * N] : 0.f);
            if (y_f32 == 1) static_cast<float*>(y)[j] = sum;
            else static_cast<uint16_t*>(y)[j] = uint16_t(bf16_bits(sum));
            continue;
        }
        if (y_f32) static_cast<float*>(y)[j] = sum;
        else static_cast<_Float16*>(y)[j] = _Float16(sum);
    }
}

// K splits of the small-batch kernel.  A workgroup (four pairs of records, one range of half slabs) needs two co-resident
// neighbours to hide its memory latency, and every workgroup beyond a whole number per CU is a second round for a few CUs: the
// split is the LARGEST for which the workgroups still fit two per CU (13824 x 5120: 108 x 4 = 432 workgroups 22.6 us, 108 x 5 = 540
// 29.9 us, 108 x 2 27.8 us), with at least four half slabs per split (the partial outputs cost 2 x 4 x M x N bytes per split).
int g_sb_waves = 0;          // tools only: aim at this many working waves instead (0: the rule above)
int sb_cu_count() {
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}
// The launch geometry of the small-batch kernel for M rows: (RP row pairs x KQ K phases) per workgroup, KS splits across gridDim.y.
//   geometry 0  RP 4, KQ 1 (round 4): four row pairs share one x tile; K is split ACROSS workgroups (rule above) and sb_reduce_kernel
//               adds the partial tiles.  What 33 - 64 rows and layers with few rows run.
//   geometry 1  RP 2, KQ 4 (round 6): layers with enough rows to fill the chip with whole-K workgroups -- no split, no workspace, ONE
//               launch (13824 x 5120: 216 workgroups of 8 working waves, 128 KiB of LDS each; 19.5 us where geometry 0 + reduce take
//               22.0, profiles/r06_small_batch.md).
// Measured and not kept (same file): RP 1 x KQ 3 with a third of geometry 0's split for layers with few rows and a long K
// (5120 x 13824: 22.9 us against 21.1 for geometry 0) -- the kernel is bound by its LDS traffic per (record, half slab), not by the
// partial tiles; ring depth 4 and non-temporal slot loads (36 us / +0.4 us).
struct SbPlan { int geo, rp, kq, KS, hps; };
int g_sb_force_geo = -1, g_sb_force_ks = 0;      // tools / tests only (pbl_debug_set_small_image_plan)
SbPlan sb_plan(const pbl_layer* L, int M) {
    const int NH = int((L->K + GI_HS - 1) / GI_HS), npairs = int((L->NRB + 1) / 2), cus = sb_cu_count();
    SbPlan p = {0, SB_WAVES, 1, 1, NH};
    int geo = (M <= 32 && npairs * 4 >= cus * 5) ? 1 : 0;          // >= 160 whole-K workgroups of two pairs on 256 CUs
    if (g_sb_force_geo >= 0 && (g_sb_force_geo == 0 || M <= 32)) geo = g_sb_force_geo > 1 ? 1 : g_sb_force_geo;
    int ks = 1;
    if (geo == 0 || g_sb_waves > 0) {
        const int cols = (npairs + SB_WAVES - 1) / SB_WAVES;
        geo = 0;
        ks = g_sb_waves > 0 ? (g_sb_waves + npairs / 2) / npairs : int((PBL_SB_WGS_PER_CU * cus * 51LL / 50) / cols);
        if (ks > NH / 4) ks = NH / 4;
    } else {
        p.rp = 2; p.kq = 4;
    }
    if (g_sb_force_ks > 0 && g_sb_force_geo >= 0) ks = g_sb_force_ks;
    if (ks > NH) ks = NH;
    if (ks < 1) ks = 1;
    p.geo = geo;
    p.hps = (NH + ks - 1) / ks;
    p.KS = (NH + p.hps - 1) / p.hps;
    return p;
}

size_t align16(size_t v) { return (v + 15) & ~size_t(15); }
size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
bool layer_ok(const pbl_layer* layer) {
    if (!layer || (layer->K & 7) || !(layer->flags & PBL_FLAG_SLABS) || !(layer->flags & PBL_FLAG_TAIL_REPEAT)) return false;
    if (layer->G < 1 || (layer->G > 1 && (layer->K % layer->G || (layer->K / layer->G) % GI_HS))) return false;
    return (layer->K + GI_HS - 1) / GI_HS <= GI_MAX_NH;
}
// where the image's parts are, from the geometry words (geom[0]: all slots in 256-byte units, geom[1]: the largest nv)
struct ImgGeom { uint64_t rtab_off, slots_off, levels_off, total; };
bool make_geom(const pbl_layer* layer, const uint32_t* geom, ImgGeom& g) {
    if (!geom || geom[1] < 1 || geom[1] > GI_NVMAX) return false;          // (0xFFFFFFFF: a slot with more than 1216 entries)
    g.rtab_off = align256(sizeof(ImgHeader) + size_t(layer->NRB) * 4);
    g.slots_off = g.rtab_off + size_t(layer->NRB) * GI_TABW * 4;
    g.levels_off = g.slots_off + uint64_t(geom[0]) * 256;
    g.total = align16(g.levels_off + size_t(layer->NRB) * layer->G * 64);
    return true;
}
size_t prep_lds(const pbl_layer* layer) {
    return size_t(2 * (GI_MAX_NH + 1) + GI_TABW) * 4 + 32 * 4 + 16 * sizeof(pbl_rowinfo) + ((size_t(layer->max_nch) + 15) & ~size_t(15));
}
// the statistics buffer of pbl_gemm_image_stats: [geom: 2 words + 8 bytes][rlen -> rbase: NRB words, padded to 16 bytes][rtab: NRB x 128 words]
size_t stats_rlen_off() { return 16; }
size_t stats_rtab_off(const pbl_layer* layer) { return 16 + align16(size_t(layer->NRB) * 4); }

}  // namespace

// Bytes of the device buffer pbl_gemm_image_stats fills (0: no image for this layer: K % 8, more than 127 half slabs, an odd group size).
extern "C" size_t pbl_gemm_image_stats_bytes(const pbl_layer* layer) {
    if (!layer_ok(layer)) return 0;
    return stats_rtab_off(layer) + size_t(layer->NRB) * GI_TABW * 4;
}

// Size the image: for every (16-row record, 128-column half slab) the number of salient entries + exceptions, the slot that holds them
// (1 - 5 KiB), every record's start -- into the device buffer stats_dev (pbl_gemm_image_stats_bytes(layer), 16-byte aligned, any
// content), which pbl_gemm_image_build reads.  Its first two words are the geometry the host needs (read them back ONCE, after this
// call's kernels: the only host synchronisation of building an image): geom[0] = all slots in 256-byte units, geom[1] = the largest
// slot in 1 KiB vectors (0xFFFFFFFF: some slot has more than 1216 entries -- no image for this layer).  Two small kernels, no x.
extern "C" int pbl_gemm_image_stats(const pbl_layer* layer, void* stats_dev, void* stream) {
    if (!layer || !layer->blob || !stats_dev) return PBL_ERR_INVALID_ARG;
    if (!layer_ok(layer)) return PBL_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(stats_dev) & 15) return PBL_ERR_MISALIGNED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* sb = static_cast<char*>(stats_dev);
    if (hipMemsetAsync(sb, 0, 16, st) != hipSuccess) return PBL_ERR_LAUNCH;
    pbl_layer lcopy = *layer;
    PrepArgs pa = {};
    pa.geom = reinterpret_cast<uint32_t*>(sb);
    pa.rlen = reinterpret_cast<uint32_t*>(sb + stats_rlen_off());
    pa.rtab = reinterpret_cast<uint32_t*>(sb + stats_rtab_off(layer));
    void* argv[] = {&lcopy, &pa};
    if (hipLaunchKernel(reinterpret_cast<const void*>(img_prep_kernel<true>), dim3(layer->NRB), dim3(GI_PREP_THREADS), argv, prep_lds(layer), st) != hipSuccess)
        return PBL_ERR_LAUNCH;
    uint32_t nrb = layer->NRB;
    void* sv[] = {&pa.geom, &pa.rlen, &nrb};
    return hipLaunchKernel(reinterpret_cast<const void*>(img_scan_kernel), dim3(1), dim3(1024), sv, 0, st) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// Bytes of the layer's GEMM image for the geometry words geom[0 .. 2) (HOST array, read back from the head of pbl_gemm_image_stats'
// buffer): 64 B header + 4 B per record + 512 B of slot table per record + the slots (1 KiB per 16 rows x 128 columns with up to 192
// entries, 2 KiB up to 448 ... 5 KiB up to 1216) + 64 B of levels per (record, group).  0: no image for this layer --
// pbl_gemm_f16_ws serves it.
extern "C" size_t pbl_gemm_image_bytes(const pbl_layer* layer, const uint32_t* geom) {
    ImgGeom g;
    if (!layer_ok(layer) || !make_geom(layer, geom, g)) return 0;
    return g.total;
}

// Build the image into `image` (>= pbl_gemm_image_bytes(layer, geom), 16-byte aligned) from the layer and the statistics buffer
// pbl_gemm_image_stats filled for it: one small kernel.  It depends on the blob only and stays valid as long as the blob is
// unchanged; the statistics buffer may be released once this call's kernel has run.
static int image_build(const pbl_layer* layer, const uint32_t* geom, const void* stats_dev, void* image, size_t image_bytes, int resid, void* stream);
extern "C" int pbl_gemm_image_build(const pbl_layer* layer, const uint32_t* geom, const void* stats_dev, void* image, size_t image_bytes, void* stream) {
    return image_build(layer, geom, stats_dev, image, image_bytes, 0, stream);
}

// The RESIDUAL image of an fp32-grid layer (round 6; the reference's fp32-only module classes, quant/quantizer.py:78,175, and QAT's
// fp32 master weights, utils.py:34-36): the same slots, tables and geometry as pbl_gemm_image_build, but every value v (levels,
// salient values fl32(sscale (q - szero)), exceptions) is replaced by fp16(4096 (v - fp16(v))).  With the ordinary image (which holds
// fp16(v)) the layer's fp32 weights are  W = W_hi + 2^-12 W_lo  up to 2^-22 |W|, so  y = x W_hi^T + 2^-12 x W_lo^T  runs on the
// hand-written fp16-tile kernels and meets the fp32 classes' 2e-5 bar -- no dense fp32 copy, no library GEMM (pbl_act_f32_join3 adds
// the terms).  Not for PBL_FLAG_SAL_F16 layers (their values ARE fp16: the residual is zero).
extern "C" int pbl_gemm_image_build_residual(const pbl_layer* layer, const uint32_t* geom, const void* stats_dev, void* image, size_t image_bytes, void* stream) {
    if (layer && (layer->flags & PBL_FLAG_SAL_F16)) return PBL_ERR_UNSUPPORTED;
    return image_build(layer, geom, stats_dev, image, image_bytes, 1, stream);
}

static int image_build(const pbl_layer* layer, const uint32_t* geom, const void* stats_dev, void* image, size_t image_bytes, int resid, void* stream) {
    if (!layer || !layer->blob || !image || !geom || !stats_dev) return PBL_ERR_INVALID_ARG;
    if (!layer_ok(layer)) return PBL_ERR_UNSUPPORTED;
    ImgGeom g;
    if (!make_geom(layer, geom, g)) return PBL_ERR_UNSUPPORTED;
    if (image_bytes < g.total) return PBL_ERR_CAPACITY;
    if ((reinterpret_cast<uintptr_t>(image) & 15) || (reinterpret_cast<uintptr_t>(stats_dev) & 15)) return PBL_ERR_MISALIGNED;
    pbl_layer lcopy = *layer;
    char* sb = const_cast<char*>(static_cast<const char*>(stats_dev));
    PrepArgs pa = {};
    pa.img = static_cast<uint8_t*>(image);
    pa.rlen = reinterpret_cast<uint32_t*>(sb + stats_rlen_off());
    pa.rtab = reinterpret_cast<uint32_t*>(sb + stats_rtab_off(layer));
    pa.rtab_off = g.rtab_off; pa.slots_off = g.slots_off; pa.levels_off = g.levels_off; pa.total = g.total;
    pa.resid = resid;
    void* argv[] = {&lcopy, &pa};
    return hipLaunchKernel(reinterpret_cast<const void*>(img_prep_kernel<false>), dim3(layer->NRB), dim3(GI_PREP_THREADS), argv, prep_lds(layer),
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

namespace {

// ---- the launch plan: cutting a thin last round off and splitting it along K ---------------------------------------------------
// The kernel's unit is a tile of 128 rows x 256 tokens over the WHOLE K, one per CU and round.  T tiles on C CUs cost ceil(T / C)
// rounds whatever T % C is: 5120 x 5120 at 2048 rows is 320 tiles = two rounds for 1.25 rounds of work, a 300-token prompt on
// 4096 x 4096 is 64 tiles on 256 CUs.  Round 4 answered with the library for such shapes; round 5 keeps them on this kernel:
// the launch is cut into a FULL part (whole rounds, tiles over the whole K, written straight to y) and a TAIL part whose tiles are
// split along K into KS work items each, so that tail tiles x KS fills the chip once; the tail's fp32 partial tiles go to a region
// buffer [KS][region] in the caller's workspace and img_reduce_kernel adds them in split order (deterministic), adds the bias,
// scales (bf16 activations) and casts.  The tail is either the last token tiles of every row tile ("token tail": the region is
// the contiguous rows [t0, M) of y) or the last row tiles of every token tile ("row tail": the columns [c0, N) of y).
// Unlike round 4's stream-K attempt (partials exchanged through agent-scope stores and flags INSIDE one launch: the exchange cost
// the workers what the split saved) the partial tiles cross a launch boundary: plain stores, plain loads.
struct ImgPlan {
    int mode;                       // 0: one launch over everything (no workspace); 1: token tail; 2: row tail
    uint32_t RT, TT;                // row tiles, token tiles of the whole problem
    uint32_t cut;                   // mode 1: token tiles [0, cut) are the full part; mode 2: row tiles [0, cut)
    int KS, hps;                    // K splits of the tail, half slabs per split
    size_t region_tok, region_col;  // the tail's region of y: tokens x columns
};
int g_img_force_mode = -1, g_img_force_cut = 0, g_img_force_ks = 0;      // tests / tools only (pbl_debug_force_gemm_plan)

ImgPlan img_plan(const pbl_layer* layer, int M) {
    ImgPlan p = {};
    p.RT = (layer->NRB + 7) / 8;
    p.TT = uint32_t((M + GI_TOK - 1) / GI_TOK);
    const int NH = int((layer->K + GI_HS - 1) / GI_HS);
    const double cus = double(sb_cu_count());
    const uint32_t T = p.RT * p.TT;
    auto region = [&](ImgPlan& q) {
        if (q.mode == 1) { q.region_tok = size_t(M) - size_t(q.cut) * GI_TOK; q.region_col = layer->N; }
        else { q.region_tok = size_t(M); q.region_col = size_t(layer->N) - size_t(q.cut) * GI_ROWS; }
    };
    if (g_img_force_mode >= 0) {
        p.mode = g_img_force_mode;
        if (p.mode) {
            p.cut = uint32_t(g_img_force_cut); p.KS = g_img_force_ks > 0 ? g_img_force_ks : 2;
            if (p.cut >= (p.mode == 1 ? p.TT : p.RT) || NH / p.KS < 2) { p.mode = 0; return p; }
            p.hps = (NH + p.KS - 1) / p.KS; p.KS = (NH + p.hps - 1) / p.hps;
            region(p);
        }
        return p;
    }
    // cost model in us (measured, profiles/r04_gemm.md / r05_gemm.md): a round costs ~2.06 us per half slab of K plus ~6 us of
    // start-up and tail per work item; the reduce reads KS fp32 regions and writes one at ~3 TB/s behind a ~4 us launch; two
    // more launches cost ~3 us.  Only the ORDER of the candidates matters.
    const double t_hs = 2.06, c0 = 6.0;
    auto rounds = [&](double items) { return items <= 0 ? 0.0 : double(uint64_t((items + cus - 1) / cus)); };
    const double unsplit = rounds(T) * (NH * t_hs + c0);
    double best = unsplit * 0.95;                       // a split has to be worth 5 % in the model (which over-prices thin one-launch rounds: call r5-10)
    for (int mode = 1; mode <= 2; ++mode) {
        const uint32_t n = mode == 1 ? p.TT : p.RT, other = mode == 1 ? p.RT : p.TT;
        for (uint32_t cut = 0; cut < n; ++cut) {
            const double full = double(cut) * other, tail = double(n - cut) * other;
            if (cut && rounds(full) * cus - full > 0.12 * cus) continue;      // the full part must be (nearly) whole rounds
            for (int ks = 2; ks <= 8 && NH / ks >= 4; ++ks) {
                const int hps = (NH + ks - 1) / ks, KS = (NH + hps - 1) / hps;
                if (NH - (KS - 1) * hps < 2) continue;                         // (no sliver of a last split)
                ImgPlan q = p;
                q.mode = mode; q.cut = cut; q.KS = KS; q.hps = hps;
                region(q);
                const double bytes = double(q.region_tok) * double(q.region_col) * (4.0 * KS + 2.0);
                const double cost = rounds(full) * (NH * t_hs + c0) + rounds(tail * KS) * (hps * t_hs + c0) + 4.0 + bytes / 3.0e6 + 3.0;
                if (cost < best) { best = cost; p = q; }
            }
        }
    }
    return p;
}

// y region = sum over the KS partial regions in split order (+ bias, x tok_scale, cast): 4 columns per thread
template <int OM>
__global__ __launch_bounds__(256) void img_reduce_kernel(const float* __restrict__ part, size_t stride, int KS, void* __restrict__ y, uint32_t ldy,
                                                         uint32_t tok0, uint32_t col0, uint32_t ntok, uint32_t ncol, const float* __restrict__ bias,
                                                         const float* __restrict__ tok_scale) {
    const size_t i = (size_t(blockIdx.x) * 256 + threadIdx.x) * 4;      // element of the region, ncol % 4 == 0 (ncol: a multiple of 128 or N % 4 == 0 checked by the host)
    if (i >= size_t(ntok) * ncol) return;
    const uint32_t t = uint32_t(i / ncol), c = uint32_t(i - size_t(t) * ncol);
    v4f sum = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KS; k0 += 8) {
        v4f v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = k0 + j < KS ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(part + size_t(k0 + j) * stride + i)) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j < KS) sum = (k0 + j) ? sum + v[j] : v[j];
    }
    v4f b = {0.f, 0.f, 0.f, 0.f};
    if (bias) b = *reinterpret_cast<const v4f*>(bias + col0 + c);
    float o[4];
    const float sc = OM == 2 ? tok_scale[tok0 + t] : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = OM == 2 ? __builtin_fmaf(sum[r], sc, b[r]) : sum[r] + b[r];
    const size_t dst = size_t(tok0 + t) * ldy + col0 + c;
    if (OM == 1) *reinterpret_cast<v4f*>(static_cast<float*>(y) + dst) = v4f{o[0], o[1], o[2], o[3]};
    else {
        uint2 pk;
        if (OM == 2) { pk.x = bf16_bits(o[0]) | (bf16_bits(o[1]) << 16); pk.y = bf16_bits(o[2]) | (bf16_bits(o[3]) << 16); }
        else { pk.x = h16(o[0]) | (h16(o[1]) << 16); pk.y = h16(o[2]) | (h16(o[3]) << 16); }
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(y) + dst) = pk;
    }
}

int img_launch(const ImgArgs& a0, int om, bool kt, hipStream_t st) {
#define GI_PICK(OM_) (a0.xf ? reinterpret_cast<const void*>(pbl_gemm_img_kernel<OM_, false, true>) \
                            : (kt ? reinterpret_cast<const void*>(pbl_gemm_img_kernel<OM_, true, false>) : reinterpret_cast<const void*>(pbl_gemm_img_kernel<OM_, false, false>)))
    const void* k = om == 1 ? GI_PICK(1) : (om == 2 ? GI_PICK(2) : GI_PICK(0));
#undef GI_PICK
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(GI_LDS)) != hipSuccess) return PBL_ERR_LAUNCH;
    ImgArgs a = a0;
    void* argv[] = {&a};
    const dim3 grid(a.nrt * a.ntt * uint32_t(a.KSn));
    if (!grid.x) return PBL_OK;
    return hipLaunchKernel(k, grid, dim3((GI_NCONS + GI_NPROD) * GW), argv, GI_LDS, st) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

}  // namespace

// tests / tools: force the launch plan of pbl_gemm_f16_image_ws (mode 0: one launch; 1: token tail; 2: row tail; cut: tiles of the
// full part; ks: K splits of the tail); mode < 0: back to the cost model
extern "C" void pbl_debug_force_gemm_plan(int mode, int cut, int ks) { g_img_force_mode = mode; g_img_force_cut = cut; g_img_force_ks = ks; }

// the plan pbl_gemm_f16_image_ws takes for M rows: out[0 .. 6) = mode (0 one launch, 1 token tail, 2 row tail), cut (tiles of the
// full part along the cut dimension), K splits of the tail, half slabs per split, region tokens, region columns
extern "C" int pbl_gemm_image_plan(const pbl_layer* layer, int M, uint64_t* out6) {
    if (!layer || !out6 || M < 1) return PBL_ERR_INVALID_ARG;
    if (!layer_ok(layer)) return PBL_ERR_UNSUPPORTED;
    const ImgPlan p = img_plan(layer, M);
    out6[0] = uint64_t(p.mode); out6[1] = p.cut; out6[2] = uint64_t(p.KS); out6[3] = uint64_t(p.hps); out6[4] = p.region_tok; out6[5] = p.region_col;
    return PBL_OK;
}

// Transient workspace pbl_gemm_f16_image_ws wants for M rows of x (0: the plan is one launch): the fp32 partial regions of the
// K-split tail.  16-byte aligned, any content, from the caller's allocator.
extern "C" size_t pbl_gemm_image_workspace_bytes(const pbl_layer* layer, int M) {
    if (!layer_ok(layer) || M < 1) return 0;
    const ImgPlan p = img_plan(layer, M);
    if (!p.mode || (p.region_col & 3)) return 0;
    return size_t(p.KS) * p.region_tok * p.region_col * sizeof(float);
}

// y[M, N] = x[M, K] . W^T (+ bias) over an image pbl_gemm_image_build made for THIS layer, with the same geometry words (any
// M >= 1).  out_dtype: PBL_DTYPE_F16 / PBL_DTYPE_F32, or PBL_DTYPE_BF16 with tok_scale [M] (device, fp32; pbl_act_bf16_prepare wrote
// it next to the fp16 copy of the bf16 activations): y[t, r] = bf16(acc[t, r] * tok_scale[t] + bias[r]).
// workspace (pbl_gemm_image_workspace_bytes(layer, M), 16-byte aligned; NULL / too small: one launch over everything, bit-identical
// to pbl_gemm_f16_ws / _prepared): with it a thin last round is cut off and split along K (see ImgPlan) -- the tiles of the full part
// keep those bits, the tail's differ by fp32 summation order (within the parity tolerance; repeatable run to run).
static int image_gemm(const pbl_layer* layer, const void* x, const void* xf, void* y, int M, int out_dtype, const float* tok_scale,
                      const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int pbl_gemm_f16_image_ws(const pbl_layer* layer, const void* x, void* y, int M, int out_dtype, const float* tok_scale,
                                     const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return image_gemm(layer, x, nullptr, y, M, out_dtype, tok_scale, image, image_bytes, geom, workspace, workspace_bytes, stream);
}

// ---- x as a fragment-major copy (round 6) -------------------------------------------------------------------------------------------
// Columns of the copy: K rounded up to a 64-column step + one step of zeros (the MFMA waves load four k-steps ahead).
static uint32_t xf_kp_of(uint32_t K) { return ((K + GI_XC - 1) / GI_XC) * GI_XC + (PBL_XF_DEPTH / 4) * GI_XC; }

// Bytes of the fragment-major copy of x [M, K] fp16 that pbl_x_to_fragments writes and pbl_gemm_f16_image_xf reads: token blocks up to
// a whole 256-token tile x (K rounded up to 64, + 64) columns x 2.
extern "C" size_t pbl_x_fragment_bytes(int M, uint32_t K) {
    if (M < 1 || K < 1) return 0;
    return size_t((M + GI_TOK - 1) / GI_TOK) * GI_TOK * size_t(xf_kp_of(K)) * 2;
}

namespace {
// one workgroup: one block of 32 tokens x up to 32 k-steps (512 columns).  Thread t: lane t & 63 of the fragment, k-steps t >> 6, + 4, ...;
// reads 16 bytes of one token row (zeros beyond M / K), writes its 16 bytes of the fragment: whole 1 KiB fragments per wave.  (The four
// waves of a workgroup read the four k-steps of a 128-byte line at the same time: the 32-byte reads meet in L1.  A variant whose
// wave-loads are eight whole lines and whose wave-stores are eight 128-byte runs measured SLOWER -- 20.2 vs 15.5 us on 2048 x 11008,
// 10.7 vs 7.4 on 2048 x 5120, call r6m: contiguous KiB stores matter more than whole-line loads.)
__global__ __launch_bounds__(256) void x_fragments_kernel(const _Float16* __restrict__ x, int M, uint32_t K, size_t ldx, _Float16* __restrict__ xf, uint32_t kp) {
    const uint32_t tb = blockIdx.x, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t tok = tb * 32u + (lane & 31u), g = lane >> 5;
    const uint32_t ks0 = blockIdx.y * 32u, nks = kp / 16u;
    char* dst = reinterpret_cast<char*>(xf) + size_t(tb) * kp * 64u + size_t(lane) * 16u;
    const bool row_ok = int(tok) < M;
    const bool vec = (K & 7u) == 0 && (ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t ks = ks0 + w + 4u * i;
        if (ks >= nks) break;
        const uint32_t col = ks * 16u + g * 8u;
        u32x4 v = {0, 0, 0, 0};
        if (row_ok && col < K) {
            const _Float16* src = x + size_t(tok) * ldx + col;
            if (vec) v = *reinterpret_cast<const u32x4*>(src);
            else {
                uint16_t h[8];
#pragma unroll
                for (int e_ = 0; e_ < 8; ++e_) h[e_] = col + uint32_t(e_) < K ? __builtin_bit_cast(uint16_t, src[e_]) : uint16_t(0);
                v = u32x4{uint32_t(h[0]) | (uint32_t(h[1]) << 16), uint32_t(h[2]) | (uint32_t(h[3]) << 16), uint32_t(h[4]) | (uint32_t(h[5]) << 16),
                          uint32_t(h[6]) | (uint32_t(h[7]) << 16)};
            }
        }
        *reinterpret_cast<u32x4*>(dst + size_t(ks) * 1024u) = v;
    }
}
}  // namespace

// x [M, K] fp16 (device, rows ldx elements apart) -> its fragment-major copy (pbl_x_fragment_bytes(M, K) bytes, 16-byte aligned): for
// every block of 32 tokens and every 16-column k-step the 1 KiB an MFMA B fragment is -- lane l of 64 holds token l & 31, columns
// 16 ks + 8 (l >> 5) .. + 7 -- so that pbl_gemm_f16_image_xf's matrix-core waves load their fragments straight from memory and no
// x tile passes through LDS.  Tokens beyond M and columns beyond K are zeros.  ONE small streaming kernel per distinct x: the q / k / v
// projections of a decoder layer share one copy, gate / up another (gptq_pb/eval_ppl_utils.py:55-64 calls them with the same tensor).
extern "C" int pbl_x_to_fragments(const void* x_f16, int M, uint32_t K, size_t ldx, void* xf, void* stream) {
    if (!x_f16 || !xf || M < 1 || K < 1 || ldx < K) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x_f16) & 1) || (reinterpret_cast<uintptr_t>(xf) & 15)) return PBL_ERR_MISALIGNED;
    const _Float16* x = static_cast<const _Float16*>(x_f16);
    _Float16* o = static_cast<_Float16*>(xf);
    uint32_t kp = xf_kp_of(K);
    const uint32_t tbs = uint32_t((M + GI_TOK - 1) / GI_TOK) * (GI_TOK / 32);
    void* argv[] = {&x, &M, &K, &ldx, &o, &kp};
    return hipLaunchKernel(reinterpret_cast<const void*>(x_fragments_kernel), dim3(tbs, (kp / 16 + 31) / 32), dim3(256), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// pbl_gemm_f16_image_ws with x given as the fragment-major copy pbl_x_to_fragments made of it (same M, same K): the same plans, the
// same results up to nothing -- every accumulator sums the same products in the same order (bit-identical to pbl_gemm_f16_image_ws).
extern "C" int pbl_gemm_f16_image_xf(const pbl_layer* layer, const void* x_fragments, void* y, int M, int out_dtype, const float* tok_scale,
                                     const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return image_gemm(layer, nullptr, x_fragments, y, M, out_dtype, tok_scale, image, image_bytes, geom, workspace, workspace_bytes, stream);
}

static int image_gemm(const pbl_layer* layer, const void* x, const void* xf, void* y, int M, int out_dtype, const float* tok_scale,
                      const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream) {
    if (!layer || !layer->blob || (!x && !xf) || !y || !image || !geom || M < 1) return PBL_ERR_INVALID_ARG;
    if (!x) x = xf;          // (alignment checks below)
    if (out_dtype != PBL_DTYPE_F16 && out_dtype != PBL_DTYPE_F32 && out_dtype != PBL_DTYPE_BF16) return PBL_ERR_INVALID_ARG;
    if ((out_dtype == PBL_DTYPE_BF16) != (tok_scale != nullptr)) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(image) & 15) ||
        (reinterpret_cast<uintptr_t>(tok_scale) & 3)) return PBL_ERR_MISALIGNED;
    if (!layer_ok(layer)) return PBL_ERR_UNSUPPORTED;
    ImgGeom g;
    if (!make_geom(layer, geom, g)) return PBL_ERR_UNSUPPORTED;
    if (image_bytes < g.total) return PBL_ERR_CAPACITY;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int NH = int((layer->K + GI_HS - 1) / GI_HS);
    const int om = out_dtype == PBL_DTYPE_F32 ? 1 : (out_dtype == PBL_DTYPE_BF16 ? 2 : 0);
    ImgArgs a;
    const uint8_t* ib = static_cast<const uint8_t*>(image);
    a.L = *layer; a.x = xf ? nullptr : static_cast<const _Float16*>(x); a.y = y; a.M = M; a.y_f32 = om == 1; a.tok_scale = tok_scale;
    a.xf = static_cast<const _Float16*>(xf); a.xf_kp = xf_kp_of(layer->K);
    a.slots = ib + g.slots_off; a.rbase = reinterpret_cast<const uint32_t*>(ib + sizeof(ImgHeader));
    a.rtab = reinterpret_cast<const uint32_t*>(ib + g.rtab_off); a.levels = reinterpret_cast<const uint32_t*>(ib + g.levels_off);
#if PBL_TRACE
    a.trace = g_img_trace;
#endif
    const bool kt = (layer->K & (GI_XC - 1)) != 0;
    ImgPlan p = img_plan(layer, M);
    const size_t need = p.mode ? size_t(p.KS) * p.region_tok * p.region_col * sizeof(float) : 0;
    // (the reduce reads the bias as 16-byte vectors: a bias that is not 16-byte aligned keeps the one-launch form -- ADVICE r5)
    if (p.mode && (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) || (p.region_col & 3) ||
                   (layer->N & (om == 1 ? 3u : 7u)) || (reinterpret_cast<uintptr_t>(layer->bias) & 15))) p.mode = 0;
    // the full part (everything when the plan is one launch): whole K, straight to y
    a.rt0 = 0; a.tt0 = 0; a.nrt = p.mode == 2 ? p.cut : p.RT; a.ntt = p.mode == 1 ? p.cut : p.TT;
    a.h0 = 0; a.nh = NH; a.hps = NH; a.KSn = 1; a.part_stride = 0; a.ldy = layer->N; a.ycol0 = 0; a.ytok0 = 0;
    int rc = img_launch(a, om, kt, st);
    if (rc != PBL_OK || !p.mode) return rc;
    // the tail: KS work items per tile, fp32 partial tiles into the region buffer (no bias, no scale: the reduce applies them)
    ImgArgs t = a;
    t.L.bias = nullptr; t.tok_scale = nullptr; t.y = workspace; t.y_f32 = 1;
    if (p.mode == 1) { t.rt0 = 0; t.nrt = p.RT; t.tt0 = p.cut; t.ntt = p.TT - p.cut; t.ycol0 = 0; t.ytok0 = int(p.cut) * GI_TOK; }
    else { t.rt0 = p.cut; t.nrt = p.RT - p.cut; t.tt0 = 0; t.ntt = p.TT; t.ycol0 = p.cut * GI_ROWS; t.ytok0 = 0; }
    t.hps = p.hps; t.KSn = p.KS; t.part_stride = p.region_tok * p.region_col; t.ldy = uint32_t(p.region_col);
    rc = img_launch(t, 1, kt, st);
    if (rc != PBL_OK) return rc;
    const float* part = static_cast<const float*>(workspace);
    size_t stride = t.part_stride;
    int KS = p.KS;
    uint32_t ldy = layer->N, tok0 = uint32_t(t.ytok0), col0 = t.ycol0, ntok = uint32_t(p.region_tok), ncol = uint32_t(p.region_col);
    const float* bias = layer->bias;
    void* rv[] = {&part, &stride, &KS, &y, &ldy, &tok0, &col0, &ntok, &ncol, &bias, &tok_scale};
    const void* rk = om == 1 ? reinterpret_cast<const void*>(img_reduce_kernel<1>) : (om == 2 ? reinterpret_cast<const void*>(img_reduce_kernel<2>)
                                                                                                : reinterpret_cast<const void*>(img_reduce_kernel<0>));
    const size_t n4 = (size_t(ntok) * ncol + 3) / 4;
    return hipLaunchKernel(rk, dim3(uint32_t((n4 + 255) / 256)), dim3(256), rv, 0, st) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// the same without a workspace: ONE launch over everything, bit-identical to pbl_gemm_f16_ws / _prepared
extern "C" int pbl_gemm_f16_image_ex(const pbl_layer* layer, const void* x, void* y, int M, int out_dtype, const float* tok_scale,
                                     const void* image, size_t image_bytes, const uint32_t* geom, void* stream) {
    return pbl_gemm_f16_image_ws(layer, x, y, M, out_dtype, tok_scale, image, image_bytes, geom, nullptr, 0, stream);
}

extern "C" int pbl_gemm_f16_image(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, const void* image, size_t image_bytes,
                                  const uint32_t* geom, void* stream) {
    return pbl_gemm_f16_image_ws(layer, x, y, M, y_f32 ? PBL_DTYPE_F32 : PBL_DTYPE_F16, nullptr, image, image_bytes, geom, nullptr, 0, stream);
}

// tuning hook (tools/): the number of waves the small-batch kernel's K split aims at
extern "C" void pbl_debug_set_small_image_waves(int n) { g_sb_waves = n > 0 ? n : 0; }
// tools / tests: force the small-batch kernel's geometry (0: RP 4 x KQ 1, 1: RP 2 x KQ 4 -- up to 32 rows) and, with ks > 0, its
// split across workgroups; geo < 0: back to the rule (sb_plan)
extern "C" void pbl_debug_set_small_image_plan(int geo, int ks) { g_sb_force_geo = geo; g_sb_force_ks = ks; }

// Transient workspace of pbl_gemm_small_image_ws for M <= 64 rows: the K splits' fp32 partial outputs (0: one split).
extern "C" size_t pbl_gemm_small_image_workspace_bytes(const pbl_layer* layer, int M) {
    if (!layer_ok(layer) || M < 1 || M > 64) return 0;
    const SbPlan p = sb_plan(layer, M);
    return p.KS > 1 ? size_t(p.KS) * M * layer->N * sizeof(float) : 0;
}

// the small-batch kernel's launches behind pbl_gemm_small_image_ws / _act (tok_scale: the _act form)
static int sb_launch(const pbl_layer* layer, const void* x, void* y, int M, int out_dtype, const float* tok_scale, const void* image, size_t image_bytes,
                     const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream) {
    if (!layer || !layer->blob || !x || !y || !image || !geom || M < 1 || M > 64) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(image) & 15)) return PBL_ERR_MISALIGNED;
    if (!layer_ok(layer) || layer->K < 16) return PBL_ERR_UNSUPPORTED;
    ImgGeom g;
    if (!make_geom(layer, geom, g)) return PBL_ERR_UNSUPPORTED;
    if (image_bytes < g.total) return PBL_ERR_CAPACITY;
    SbArgs a;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint8_t* ib = static_cast<const uint8_t*>(image);
    a.L = *layer; a.x = static_cast<const _Float16*>(x); a.y = y; a.M = M; a.y_f32 = out_dtype == PBL_DTYPE_F32;
    a.slots = ib + g.slots_off; a.rbase = reinterpret_cast<const uint32_t*>(ib + sizeof(ImgHeader));
    a.rtab = reinterpret_cast<const uint32_t*>(ib + g.rtab_off); a.levels = reinterpret_cast<const uint32_t*>(ib + g.levels_off);
    SbPlan pl = sb_plan(layer, M);
    const uint32_t NH = (layer->K + GI_HS - 1) / GI_HS;
    // (no workspace: geometry 0 falls back to ONE split per layer -- slow for layers with few rows; the K-phase geometries then cover
    // the whole K inside their workgroups)
    if (pl.KS > 1 && (!workspace || workspace_bytes < size_t(pl.KS) * M * layer->N * sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15))) { pl.KS = 1; pl.hps = int(NH); }
    a.KS = pl.KS; a.hps = pl.hps;
    if (tok_scale) {                                         // scaled activations: scale and bias belong to the reduce
        if (a.KS == 1) return PBL_ERR_UNSUPPORTED;
        a.L.bias = nullptr;
    }
    a.part = a.KS > 1 ? static_cast<float*>(workspace) : nullptr;
    const uint32_t nvk = geom[1];
    const bool kt = (layer->K & (GI_HS - 1)) != 0;
    const int ntb = M > 32 ? 2 : 1;
#define SB_PICK3(NV_, KT_, NTB_) (pl.geo == 1 ? reinterpret_cast<const void*>(pbl_sb_img_kernel<NV_, KT_, 1, 2, 4>) \
                                              : reinterpret_cast<const void*>(pbl_sb_img_kernel<NV_, KT_, NTB_, SB_WAVES, 1>))
#define SB_PICK2(NV_, KT_) (ntb == 2 ? SB_PICK3(NV_, KT_, 2) : SB_PICK3(NV_, KT_, 1))
#define SB_PICK(NV_) (kt ? SB_PICK2(NV_, true) : SB_PICK2(NV_, false))
    const void* k = nvk == 1 ? SB_PICK(1) : nvk == 2 ? SB_PICK(2) : nvk == 3 ? SB_PICK(3) : nvk == 4 ? SB_PICK(4) : SB_PICK(5);
#undef SB_PICK
#undef SB_PICK2
#undef SB_PICK3
    void* argv[] = {&a};
    const uint32_t npairs = (layer->NRB + 1) / 2;
    if (hipLaunchKernel(k, dim3((npairs + pl.rp - 1) / pl.rp, uint32_t(a.KS)), dim3((pl.rp * pl.kq + 1) * GW), argv, 0, st) != hipSuccess) return PBL_ERR_LAUNCH;
    if (a.KS > 1) {
        const float* part = a.part;
        size_t MN = size_t(M) * layer->N;
        int KS = a.KS, om = tok_scale ? (out_dtype == PBL_DTYPE_F32 ? 1 : 2) : a.y_f32;
        const float* bias = layer->bias;
        uint32_t N = layer->N;
        void* rv[] = {&part, &y, &KS, &MN, &om, &tok_scale, &bias, &N};
        const void* rk = tok_scale ? reinterpret_cast<const void*>(sb_reduce_kernel<true>) : reinterpret_cast<const void*>(sb_reduce_kernel<false>);
        if (hipLaunchKernel(rk, dim3(uint32_t((MN + 1023) / 1024)), dim3(256), rv, 0, st) != hipSuccess) return PBL_ERR_LAUNCH;
    }
    return PBL_OK;
}

// y[M, N] = x[M, K] . W^T (+ bias) for 1 <= M <= 64 rows over the GEMM image (the same image, the same geometry words as
// pbl_gemm_f16_image).  `workspace` (pbl_gemm_small_image_workspace_bytes(layer, M), 16-byte aligned; any content) holds the K
// splits' partial outputs; NULL / too small: one split (slow for layers with few rows).  The same numbers as the other kernels up
// to fp32 summation order (within the parity tolerance of tests/test_gpu_gemm.py); repeatable run to run.
extern "C" int pbl_gemm_small_image_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, const void* image, size_t image_bytes,
                                       const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream) {
    return sb_launch(layer, x, y, M, y_f32 ? PBL_DTYPE_F32 : PBL_DTYPE_F16, nullptr, image, image_bytes, geom, workspace, workspace_bytes, stream);
}

// The same for activations scaled per token by pbl_act_bf16_prepare: y = cast(acc * tok_scale[token] + bias) with out_dtype
// PBL_DTYPE_BF16 / PBL_DTYPE_F32 -- pbl_act_finish folded into the K splits' reduce, so two launches after the prepare instead of
// three.  PBL_ERR_UNSUPPORTED (nothing launched) when the layer runs as ONE split (no reduce to fold into: small workspace, or a
// layer with enough rows to fill the device alone): the caller then runs pbl_gemm_small_image_ws (fp32) + pbl_act_finish.
extern "C" int pbl_gemm_small_image_act(const pbl_layer* layer, const void* x_f16, void* y, int M, int out_dtype, const float* tok_scale,
                                        const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream) {
    if (!tok_scale || (out_dtype != PBL_DTYPE_F32 && out_dtype != PBL_DTYPE_BF16)) return PBL_ERR_INVALID_ARG;
    return sb_launch(layer, x_f16, y, M, out_dtype, tok_scale, image, image_bytes, geom, workspace, workspace_bytes, stream);
}
