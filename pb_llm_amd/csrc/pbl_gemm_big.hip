// pbl_gemm_big.hip -- GEMM regime (more than 32 tokens: prefill, large batches) straight from the PBL1 packed format.
// Replaces F.linear(x, W_fq, b) over the dense fp16 fake-quant weight (gptq_pb/eval_ppl_utils.py:55-64 and
// evaluate.py:126-145: the reference's perplexity loops call every nn.Linear with seq 2048 rows) WITHOUT the dense weight
// ever existing in HBM.
//
// Every weight is rebuilt in LDS as the fp16 number a dense fp16 copy of the layer would hold -- one of the row's
// (row-group's) two levels, the fp16-rounded salient value, or an explicit exception -- and fed to
// v_mfma_f32_32x32x16_f16: the arithmetic is that of an fp16 GEMM with fp32 accumulation on pbl_unpack_dev's output, for
// every layer kind (fp16-checkpoint and fp32-grid layers, with or without column groups; fp16 or fp32 result).
//
// Round 3 rebuild (round 2's kernel: 16 waves, x through registers + ds_write, 4x4 accumulator tiles per wave: 570 TFLOP/s,
// LDS ~90 % busy, producers' ~250 VALU per sub-step next to the MFMA waves).  What changed and why:
//   * Workgroup = 8 waves, SPECIALISED, one consumer + one producer per SIMD.  Tile = 8 records (128 rows) x 256 tokens.
//   * Consumer wave c owns ALL 128 rows x tokens [64 c, 64 c + 64): 8 accumulator tiles of 32 x 32 (128 registers).  Per
//     64-column sub-step it reads 16 A + 8 B fragments (ds_read_b128) for 32 MFMAs: 96 KB of LDS reads per sub-step and
//     workgroup instead of 128 KB, and the A tile is read by 4 waves instead of 8.
//   * x never passes through a VGPR or a ds_write: each consumer stages the 64 tokens ONLY IT reads with
//     buffer_load_dwordx4 ... lds (LDS-DMA, 8 x 1 KiB per sub-step, whole 128-byte lines) into a PRIVATE ring of three
//     8 KiB slots; no other wave touches them, so x needs no workgroup barrier at all -- a counted vmcnt and the wave's own
//     lgkmcnt order it.  The LDS image is lane-linear (the DMA's rule), the XOR swizzle that makes the b128 fragment reads
//     conflict-free is applied to the SOURCE address.  Out-of-range tokens read zeros through the buffer descriptor.
//   * Producer wave p expands records 2p, 2p+1 of the next 128-column half slab into As[2][128][128] fp16 (XOR-swizzled
//     16-byte units, no padding).  Sign plane: (d >> pos) & 0x00010001 puts the bits of two columns into the two halves,
//     ONE v_pk_mad_u16 with op_sel turns them into {hi, lo} bit patterns (lo + bit * (hi - lo) mod 2^16; the row's
//     (hi - lo : lo) pair sits in one SGPR), one XOR forms the swizzled address: 4 VALU per weight pair and ds_write_b32,
//     all 64 lanes busy.  Salients: the packer's slab index, four lanes per chunk, entries outside the half slab masked off.
//   * Sync: ONE workgroup barrier per half slab (2 sub-steps).  Consumers pass it right after their last fragment read of
//     the stage was issued and has returned, i.e. one k-step BEFORE the stage's last MFMAs, so those MFMAs cover the first
//     fragment reads of the next stage; producers pass it when the next stage is complete and then have a whole half slab
//     of time for the one after.
//   * Epilogue per consumer wave through its own (now idle) ring: accumulators (+ bias) -> [token][row] -> 16-byte stores.
// LDS: 2 x 32 KiB (A) + 4 x 24 KiB (x rings) = 160 KiB: one workgroup per CU, by design.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/pbl.h"

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define GW 64
#ifndef NCONS
#define NCONS 8                   // consumer (MFMA) waves: 256 / NCONS tokens each.  8 = two per SIMD: one's waits are the other's MFMAs
#endif
#define GB_TPC (256 / NCONS)      // tokens per consumer wave (64 / 32)
#define GB_TT (GB_TPC / 32)       // 32-token accumulator tiles per consumer (2 / 1)
#define GB_PQ (GB_TPC / 8)        // 1 KiB DMA pieces per x slot (8 / 4)
#define NPROD 4                   // producer (expand) waves of the in-kernel-decode build: 2 records each
#ifndef NPROD_LIST
#define NPROD_LIST 4              // ... of LIST mode: 2 records each (8 x 1 record measured equal, r3p: the producers are not the critical path)
#endif
#define NREC 8                    // records per workgroup tile
#define GB_ROWS (NREC * 16)
#define GB_TOK 256
#define GB_HS 128                 // columns per A stage (half a slab)
#define GB_XC 64                  // columns per x slot (sub-step)
#define GB_XSLOTS 3
#define GB_AS_STAGE (GB_ROWS * GB_HS * 2)          // 32768 B
#define GB_XSLOT_BYTES (GB_TPC * GB_XC * 2)         // 8192 / 4096 B: the consumer's tokens x 64 columns
#define GB_XRING_BYTES (GB_XSLOTS * GB_XSLOT_BYTES)
#define GB_X_OFF (2 * GB_AS_STAGE)
#define GB_LDS (GB_X_OFF + NCONS * GB_XRING_BYTES)  // 163840 B
// performance-analysis hook (tools/build_variant.sh): bit 0 no expansion in the loop, 1 no x staging in the loop, 2 no MFMA,
// 3 no sign-plane expansion, 4 no salient overlay, 5 producers request nothing either (consumer loop alone), 6 no fragment
// reads in the loop, 7 no barriers in the loop, 8 no result stores, 9 no epilogue.  0 in every
// shipped build (results are wrong otherwise).
#ifndef PBL_GEMM_ABLATE
#define PBL_GEMM_ABLATE 0
#endif
// consumers at raised wave priority (measured variants: tools/build_variant.sh)
#ifndef PBL_GEMM_PRIO
#define PBL_GEMM_PRIO 0
#endif
// producers at raised wave priority: their ~200 VALU / LDS instructions per half slab otherwise get the issue slots the MFMA
// waves of the same SIMD leave over, and the stage (hence the barrier, hence the MFMA waves) is late
#ifndef PBL_GEMM_PPRIO
#define PBL_GEMM_PPRIO 1
#endif
// experiments on the arbitration between the two MFMA waves of a SIMD (waves c and c + NCONS / 2; oldest first by default):
// PBL_GEMM_CPRIO_YOUNG: static priority for the younger half; PBL_GEMM_ALT_PRIO: the priority flips every k-step
#ifndef PBL_GEMM_CPRIO_YOUNG
#define PBL_GEMM_CPRIO_YOUNG 0
#endif
#ifndef PBL_GEMM_ALT_PRIO
#define PBL_GEMM_ALT_PRIO 0
#endif

// timeline probe (tools/trace_gemm.py, build/libpbl_trace.so only): every wave stamps s_memrealtime (100 MHz) and s_memtime
// (shader clock) at entry / loop start / loop end / exit and adds up the shader cycles it spends parked at the workgroup
// barrier and at counted vmcnt waits.  0 in the shipped build.
#ifndef PBL_TRACE
#define PBL_TRACE 0
#endif
#if PBL_TRACE
static uint64_t* g_gemm_trace = nullptr;
extern "C" void pbl_debug_trace_gemm(void* p) { g_gemm_trace = static_cast<uint64_t*>(p); }
#define TR_STAMP(slot) do { if (tr && lane == 0) { tr[2 * (slot)] = __builtin_amdgcn_s_memrealtime(); tr[2 * (slot) + 1] = __builtin_readcyclecounter(); } } while (0)
#define TR_T0() const uint64_t tr_t0 = __builtin_readcyclecounter()
#define TR_ADD(acc) acc += __builtin_readcyclecounter() - tr_t0
#else
#define TR_STAMP(slot) do {} while (0)
#define TR_T0() do {} while (0)
#define TR_ADD(acc) do {} while (0)
#endif

namespace {

__device__ __forceinline__ _Float16 round_f16_twice(float prod) {
    asm volatile("" : "+v"(prod));   // keep the fp32 product: an fp16-checkpoint value is double rounded
    return _Float16(prod);
}
__device__ __forceinline__ uint32_t h16(float v) { return uint32_t(__builtin_bit_cast(uint16_t, _Float16(v))); }

struct GemmArgs {
    pbl_layer L;
    const _Float16* x;      // [M, K]
    void* y;                // [M, N] fp16 / fp32
    int M, y_f32;
    // LIST mode (pbl_gemm_f16_ws): the salient entries of every (record, half slab) as ready-to-store words, built once per
    // call by pbl_gemm_prep_kernel
    const uint32_t* ofs;    // [NRB][ofs_stride]: entry ranges per half slab, ofs[rb][h] .. ofs[rb][h + 1]
    const uint32_t* lst;    // [NRB][cap]: (byte offset in the record's 16 x 256-byte stage image << 16) | fp16 value
    uint32_t ofs_stride, cap;
#if PBL_TRACE
    uint64_t* trace;        // [workgroup][16 waves][16] (timeline probe)
#endif
};

struct Seq { int fb, fn, tb, tn; };            // the row's full / tail chunks that overlap a slab: first index, count

// a quarter of a salient chunk as one lane holds it: as requested (all 16 byte steps for the prefix, its 4 codes, the first
// column) and as the expansion uses it (its own 4 byte steps, its 4 codes, the byte offset of its first entry minus the step)
struct ChunkQ {
    u32x4 d4;
    uint32_t q;
    int col0;               // < 0: no chunk
};
struct QuarterQ {
    uint32_t dd, q, off0;   // off0: byte offset in a 2 K-byte fp16 row of the entry BEFORE this quarter's first (0x40000000: no chunk)
};

#define GB_PRE 3            // chunk rounds of a slab kept in registers (more rounds are fetched when they are needed; 3 keeps the 12-wave build free of spills)

// the chunks of one record that overlap one 256-column slab, as a producer lane holds them
struct SlabData {
    Seq sq;
    ChunkQ c[GB_PRE];
};
struct SlabReady {          // the same slab once its requests have landed: 3 registers per round instead of 6
    Seq sq;
    QuarterQ c[GB_PRE];
};

// Everything a producer wave keeps per record (two of these per wave).  Data for the expansion two stages ahead is
// requested right after a stage has been written: by the time it is needed a whole consumer half slab has passed.
struct Rec {
    const uint16_t* col0p;
    const u32x4* deltap;
    const uint32_t* codew;  // codes as dwords: chunk c, quarter s at [4 c + s]
    const uint2* exc;
    const float2* ghl;      // G > 1: per (row, group) levels
    const uint32_t* tabrow; // slab-index row of this lane's row (lane >> 2)
    const uint32_t* tile_dw;
    const pbl_rowparams* params;
    int nfull, nexc;
    pbl_rowinfo ri;         // this lane's row (lane >> 2)
    float ss, sz;           // ... and its code grid
    uint32_t tp, tc, tn;    // slab-index entries of slabs s - 1, s, s + 1, s = the next slab to be requested
    SlabReady sd;           // the slab being expanded
    SlabData sdn;           // the next one (requested 1.5 slabs ahead)
    uint32_t dcur, dnxt;    // sign-plane dwords of the next expansion and the one after
    uint32_t hl[16];        // wave uniform: (hi - lo) mod 2^16 : lo, fp16 bit patterns, of the 16 rows (current column group)
};

__device__ __forceinline__ Seq seq_of(uint32_t pe, uint32_t e) {
    Seq q;
    q.fb = int(PBL_SLAB_FE(pe)) - int(PBL_SLAB_FBACK(e)); q.fn = int(PBL_SLAB_FE(e)) - q.fb;
    q.tb = int(PBL_SLAB_TE(pe)) - int(PBL_SLAB_TBACK(e)); q.tn = int(PBL_SLAB_TE(e)) - q.tb;
    return q;
}
// round j of the row's slab sequence (its full chunks, then its tail chunks): the same chunk for the 4 lanes of a row
__device__ __forceinline__ ChunkQ load_chunkq(const Rec& R, const Seq& sq, int j, int sub) {
    ChunkQ r;
    r.col0 = -1; r.d4 = u32x4{0, 0, 0, 0}; r.q = 0;
    int c = -1;
    if (j < sq.fn) c = int(R.ri.start) + sq.fb + j;
    else if (j - sq.fn < sq.tn) c = R.nfull + int(R.ri.tailidx) + sq.tb + (j - sq.fn);
    if (c >= 0) { r.d4 = R.deltap[c]; r.q = R.codew[4 * c + sub]; r.col0 = int(R.col0p[c]); }
    return r;
}
template <typename REC>
__device__ __forceinline__ uint32_t load_dw(const REC& R, int h, int P) {      // dword (h & 3) of panel h >> 2 for this lane
    return (h >> 2) < P ? __builtin_nontemporal_load(R.tile_dw + size_t(h >> 2) * 256 + (h & 3)) : 0u;
}
// request the chunks of slab sn (its first GB_PRE rounds); the slab-index entry of the slab after it rides along
__device__ __forceinline__ void request_slab(Rec& R, SlabData& d, int sn, int NS, int sub) {
    d.sq = seq_of(R.tp, R.tc);                                   // (entries read a slab earlier: no dependent wait here)
    if (sn >= NS) { d.sq.fn = 0; d.sq.tn = 0; }
#pragma unroll
    for (int j = 0; j < GB_PRE; ++j) d.c[j] = load_chunkq(R, d.sq, j, sub);
    R.tp = R.tc; R.tc = R.tn;
    R.tn = sn + 2 < NS ? R.tabrow[sn + 2] : 0u;
}
// the 16 level pairs of column group g as (hi - lo : lo) bit patterns in SGPRs
__device__ __forceinline__ void load_levels(Rec& R, uint32_t G, uint32_t g, int lane) {
    float hi, lo;
    if (G > 1) { const float2 v = R.ghl[size_t(lane & 15) * G + g]; hi = v.x; lo = v.y; }
    else { const pbl_rowparams pr = R.params[lane & 15]; hi = pr.hi; lo = pr.lo; }
    const uint32_t hh = h16(hi), ll = h16(lo);
    const uint32_t w = (((hh - ll) & 0xFFFFu) << 16) | ll;
#pragma unroll
    for (int r = 0; r < 16; ++r) R.hl[r] = __builtin_amdgcn_readlane(w, r);
}

// ds_write_b16 of the lanes whose byte offset lies inside the row.  Plain C++ (`if (off < 256) store`) compiles to the same
// compare / saveexec / store / restore PLUS a branch over the store (LLVM skips DS instructions under an empty EXEC), and that
// branch ends the basic block: one producer wave then runs every entry's dependent chain in sequence.  Without the branch the
// whole overlay of a record is one block and the compiler interleaves the chains of all its entries.
__device__ __forceinline__ void lds_store_b16_inrow(uint32_t addr, uint32_t val, uint32_t off) {
    uint64_t sv;
    asm volatile("v_cmp_gt_u32_e32 vcc, 0x100, %1\n\ts_and_saveexec_b64 %0, vcc\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "v"(off), "v"(addr), "v"(val) : "vcc", "memory");
}

// what a lane needs of its chunk quarter, computed ONCE per slab (both half slabs use it): its own four byte steps and the
// byte offset (in the whole fp16 row) its running sum starts from
__device__ __forceinline__ QuarterQ finalize_q(const ChunkQ& c, int sub) {
    // byte sum of the deltas that precede this quarter (deltas are stored doubled = byte steps in an fp16 row)
    uint32_t pre = 0;
    pre = sub > 0 ? __builtin_amdgcn_sad_u8(c.d4[0], 0u, pre) : pre;
    pre = sub > 1 ? __builtin_amdgcn_sad_u8(c.d4[1], 0u, pre) : pre;
    pre = sub > 2 ? __builtin_amdgcn_sad_u8(c.d4[2], 0u, pre) : pre;
    // this quarter's four steps, picked with lane masks (a ?: chain compiles to divergent branches that end the basic block)
    const uint32_t m0 = sub == 0 ? ~0u : 0u, m1 = sub == 1 ? ~0u : 0u, m2 = sub == 2 ? ~0u : 0u, m3 = sub == 3 ? ~0u : 0u;
    QuarterQ r;
    r.dd = (c.d4[0] & m0) | (c.d4[1] & m1) | (c.d4[2] & m2) | (c.d4[3] & m3);
    r.q = c.q;
    r.off0 = c.col0 < 0 ? 0x40000000u : uint32_t(2 * c.col0) + pre;
    return r;
}

// entries 4 sub .. 4 sub + 3 of one chunk -> the tile row at LDS byte address rowaddr (a multiple of 256), half slab
// starting at byte cb2 of the fp16 row.  Tail padding repeats the last entry (PBL_FLAG_TAIL_REPEAT): a writer needs no count.
__device__ __forceinline__ void scatter_q(const QuarterQ& c, uint32_t cb2, uint32_t rowaddr, uint32_t swz, float ss, float sz) {
    uint32_t off = c.off0 - cb2;        // wraps (>= 256) for entries left of the half slab; "no chunk" stays far out of range
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        off += (c.dd >> (8 * e)) & 0xFFu;
        const float qf = float((c.q >> (8 * e)) & 0xFFu);
        const uint32_t v = uint32_t(__builtin_bit_cast(uint16_t, round_f16_twice(ss * (qf - sz))));
        lds_store_b16_inrow((off ^ swz) + rowaddr, v, off);
    }
}

// sign plane of one half slab of record R -> rows of the stage at LDS byte address recaddr (a multiple of 4096): lane l
// holds columns 2l, 2l+1 of all 16 rows in ONE dword (include/pbl.h: bit 16 e + pos)
template <bool TAIL, typename REC>
__device__ __forceinline__ void expand_sign(const REC& R, uint32_t d, char* smem, uint32_t recaddr, int lane, int kpairs /* valid column pairs */) {
    // unit (l >> 2) ^ row, dword l & 3:  recaddr + 256 row + (((l >> 2) ^ row) << 4) + 4 (l & 3)  ==  v ^ (0x110 row)
    const uint32_t v = recaddr + (uint32_t(lane >> 2) << 4) + (uint32_t(lane & 3) << 2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pos = r < 8 ? r + 8 : r - 8;
        const uint32_t m = (d >> pos) & 0x00010001u;
        uint32_t val;
        // per half: lo + bit * (hi - lo)  (mod 2^16): src1 = the pair's high half, src2 = its low half, for both lanes
        asm("v_pk_mad_u16 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(val) : "v"(m), "s"(R.hl[r]));
        if (TAIL && lane >= kpairs) val = 0;                          // columns beyond K (only the layer's last half slab)
        *reinterpret_cast<uint32_t*>(smem + (v ^ uint32_t(0x110 * r))) = val;
    }
    asm volatile("" ::: "memory");
}

// salients + exceptions of half slab h of record R over the plane the same wave has just written (LDS keeps a wave's
// accesses in order)
__device__ __forceinline__ void expand_sal(Rec& R, int h, char* smem, uint32_t recaddr, int lane) {
    const int rho = lane >> 2, sub = lane & 3;
    const uint32_t cb2 = uint32_t(h) * (2 * GB_HS);
    const uint32_t rowaddr = recaddr + uint32_t(rho) * 256u, swz = uint32_t(rho) << 4;
    const int n = R.sd.sq.fn + R.sd.sq.tn;
    // a round is skipped when NO row of the record has a chunk in it (wave uniform): at 5 % salients most slabs need 2 rounds
#pragma unroll
    for (int j = 0; j < GB_PRE; ++j)
        if (__any(j < n)) scatter_q(R.sd.c[j], cb2, rowaddr, swz, R.ss, R.sz);
    for (int j = GB_PRE; __any(j < n); ++j)
        scatter_q(finalize_q(load_chunkq(R, R.sd.sq, j, sub), sub), cb2, rowaddr, swz, R.ss, R.sz);
    asm volatile("" ::: "memory");
    for (int k = lane; k < R.nexc; k += GW) {                   // explicit values last
        const uint2 ex = R.exc[k];
        const uint32_t col = (ex.x & 0xFFFFu) - uint32_t(h * GB_HS), row = ex.x >> 16;
        if (col < uint32_t(GB_HS))
            *reinterpret_cast<uint16_t*>(smem + recaddr + row * 256u + ((2u * col) ^ (row << 4))) =
                __builtin_bit_cast(uint16_t, _Float16(__builtin_bit_cast(float, ex.y)));
    }
    asm volatile("" ::: "memory");
}

// the requested slab becomes the current one (its loads were issued 1.5 slabs ago)
__device__ __forceinline__ void adopt_slab(Rec& R, int sub) {
    R.sd.sq = R.sdn.sq;
#pragma unroll
    for (int j = 0; j < GB_PRE; ++j) R.sd.c[j] = finalize_q(R.sdn.c[j], sub);
}

__device__ __forceinline__ void init_rec(Rec& R, const pbl_layer& L, uint32_t rb, int lane) {
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    const uint8_t* rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    R.nfull = __builtin_amdgcn_readfirstlane(info.y);
    const int ntail = __builtin_amdgcn_readfirstlane(info.z);
    R.nexc = __builtin_amdgcn_readfirstlane(info.w);
    const uint32_t nchu = uint32_t(R.nfull + ntail);
    const int NS = int((L.K + PBL_SLAB_COLS - 1) / PBL_SLAB_COLS);
    const bool has_crow = (L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16)) != 0;
    const uint32_t tiles_off = PBL_TILES_OFF(L.G);
    const uint8_t* sal = rec + tiles_off + L.P * 1024u;
    R.col0p = reinterpret_cast<const uint16_t*>(sal);
    R.deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    R.codew = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_CODE_OFF(nchu));
    R.exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));
    const uint32_t* slabtab = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_SLAB_OFF(nchu, uint32_t(ntail), uint32_t(R.nexc), has_crow));
    R.tabrow = slabtab + (lane >> 2) * NS;
    R.tile_dw = reinterpret_cast<const uint32_t*>(rec + tiles_off) + lane * 4;
    R.params = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
    R.ghl = reinterpret_cast<const float2*>(rec + PBL_REC_GHL_OFF);
    R.ri = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF)[lane >> 2];
    const pbl_rowparams ps = R.params[lane >> 2];
    R.ss = ps.sscale; R.sz = ps.szero;
    R.tp = 0; R.tc = R.tabrow[0]; R.tn = NS > 1 ? R.tabrow[1] : 0u;
    request_slab(R, R.sdn, 0, NS, lane & 3);                     // slab 0 (stages 0, 1), then slab 1 (stages 2, 3)
    adopt_slab(R, lane & 3);
    request_slab(R, R.sdn, 1, NS, lane & 3);
    R.dcur = load_dw(R, 0, int(L.P));
    R.dnxt = load_dw(R, 1, int(L.P));
    load_levels(R, L.G, 0, lane);
}

// ---- LIST mode ---------------------------------------------------------------------------------------------------------
// A workgroup of the GEMM kernel decodes a record's salient chunks once per token tile and per half slab; at seq 2048 that is
// 8 x 2 decodes of the same chunk, by ONE producer wave whose in-order stream is the kernel's critical path (measured: 4096^2,
// 5 % salients: 102 us with the decode in the loop, 77 us without it).  So calls with more than one token tile decode ONCE, in
// a small kernel ahead of the GEMM: per (record, half slab) the entries become 4-byte words {offset in the stage image : fp16
// value} in a transient workspace (4 B per entry; the dense weight is 2 B per WEIGHT), and the GEMM's producers only copy them.
// One workgroup per record: count per half slab (LDS atomics), scan, fill.  Entry order inside a half slab is arbitrary:
// positions are distinct (tail padding repeats an entry with the same value), so the image is deterministic.
#define GB_LIST_MAX_NH 127          // entry ranges live in two registers per lane (NH + 1 <= 128): K <= 16256
#define GB_LPF 4                    // entry words per lane requested ahead: 256 entries of a (record, half slab) -- 12.5 % salients,
                                    // or the hot columns of a hessian mask (16 rows x 16 salient columns) -- never wait in the loop

#define GB_PREP_THREADS 1024
__global__ __launch_bounds__(GB_PREP_THREADS) void pbl_gemm_prep_kernel(pbl_layer L, uint32_t* __restrict__ ofs, uint32_t* __restrict__ lst,
                                                                         uint32_t ofs_stride, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    // LDS: cnt[128] cur[128] (u32), ss[16] sz[16] (f32), rowinfo[16], crow[nch] (u8: the row each chunk belongs to)
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem_p);
    uint32_t* s_cur = s_cnt + (GB_LIST_MAX_NH + 1);
    float* s_ss = reinterpret_cast<float*>(s_cur + (GB_LIST_MAX_NH + 1));
    float* s_sz = s_ss + 16;
    pbl_rowinfo* s_ri = reinterpret_cast<pbl_rowinfo*>(s_sz + 16);
    uint8_t* s_crow = reinterpret_cast<uint8_t*>(s_ri + 16);
    const uint32_t rb = blockIdx.x;
    const int tid = threadIdx.x;
    const int NH = int((L.K + GB_HS - 1) / GB_HS);
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    const uint8_t* rec = blob + size_t(info.x) * 16;
    const int nfull = int(info.y), ntail = int(info.z), nexc = int(info.w), nch = nfull + ntail;
    const uint32_t nchu = uint32_t(nch);
    const bool has_crow = (L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16)) != 0;
    const uint8_t* sal = rec + PBL_TILES_OFF(L.G) + L.P * 1024u;
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const uint32_t* codew = reinterpret_cast<const uint32_t*>(sal + PBL_SAL_CODE_OFF(nchu));
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));
    // this thread's first unit (chunk u >> 2, quarter u & 3): requested before anything else is waited for
    const int u0 = tid;
    u32x4 d4 = {0, 0, 0, 0};
    uint32_t q0 = 0, c00 = 0;
    if (u0 < 4 * nch) { d4 = deltap[u0 >> 2]; q0 = codew[u0]; c00 = col0p[u0 >> 2]; }
    if (tid < 16) {
        s_ri[tid] = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF)[tid];
        const pbl_rowparams ps = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF)[tid];
        s_ss[tid] = ps.sscale; s_sz[tid] = ps.szero;
    }
    for (int h = tid; h <= NH; h += GB_PREP_THREADS) { s_cnt[h] = 0; s_cur[h] = 0; }
    __syncthreads();
    for (int r = 0; r < 16; ++r) {                           // chunk -> row (rowinfo: full chunks [start, +nfull), tails [tailidx, +ntail))
        const pbl_rowinfo ri = s_ri[r];
        for (int k = tid; k < int(ri.nfull) + int(ri.ntail); k += GB_PREP_THREADS)
            s_crow[k < int(ri.nfull) ? int(ri.start) + k : nfull + int(ri.tailidx) + (k - int(ri.nfull))] = uint8_t(r);
    }
    __syncthreads();
    uint32_t* out = lst + size_t(rb) * cap;
    // pass 0 counts, pass 1 writes; a quarter's entries are column sorted, so runs inside one half slab share one atomic
    for (int pass = 0; pass < 2; ++pass) {
        for (int u = tid; u < 4 * nch; u += GB_PREP_THREADS) {
            const int c = u >> 2, sub = u & 3;
            u32x4 dv = d4; uint32_t q = q0, cc = c00;
            if (u != u0) { dv = deltap[c]; q = codew[u]; cc = col0p[c]; }       // (more than 256 chunks in the record)
            const uint32_t row = s_crow[c];
            uint32_t pre = 0;
            pre = sub > 0 ? __builtin_amdgcn_sad_u8(dv[0], 0u, pre) : pre;
            pre = sub > 1 ? __builtin_amdgcn_sad_u8(dv[1], 0u, pre) : pre;
            pre = sub > 2 ? __builtin_amdgcn_sad_u8(dv[2], 0u, pre) : pre;
            const uint32_t dd = sub == 0 ? dv[0] : (sub == 1 ? dv[1] : (sub == 2 ? dv[2] : dv[3]));
            uint32_t off[4];
            off[0] = 2u * cc + pre + (dd & 0xFFu);           // byte offsets in the fp16 row
            off[1] = off[0] + ((dd >> 8) & 0xFFu); off[2] = off[1] + ((dd >> 16) & 0xFFu); off[3] = off[2] + (dd >> 24);
            const float ss = s_ss[row], sz = s_sz[row];
            uint32_t pos = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t h = off[e] >> 8;
                const bool first = e == 0 || h != (off[e - 1] >> 8);
                if (first) {                                 // entries e .. of this quarter in half slab h
                    uint32_t run = 1;
#pragma unroll
                    for (int f = e + 1; f < 4; ++f) run += (off[f] >> 8) == h ? 1u : 0u;
                    if (pass == 0) atomicAdd(&s_cnt[h], run);
                    else pos = s_cnt[h] + atomicAdd(&s_cur[h], run);
                }
                if (pass == 1) {
                    const float qf = float((q >> (8 * e)) & 0xFFu);
                    const uint32_t v = uint32_t(__builtin_bit_cast(uint16_t, round_f16_twice(ss * (qf - sz))));
                    out[pos++] = ((row * 256u + ((off[e] & 0xFFu) ^ (row << 4))) << 16) | v;
                }
            }
        }
        for (int k = tid; k < nexc; k += GB_PREP_THREADS) {
            const uint2 ex = exc[k];
            const uint32_t col = ex.x & 0xFFFFu, row = ex.x >> 16, h = col >> 7, o = (2u * col) & 0xFFu;
            if (pass == 0) atomicAdd(&s_cnt[h], 1u);
            else {
                const uint32_t v = uint32_t(__builtin_bit_cast(uint16_t, _Float16(__builtin_bit_cast(float, ex.y))));
                out[s_cnt[h] + atomicAdd(&s_cur[h], 1u)] = ((row * 256u + (o ^ (row << 4))) << 16) | v;
            }
        }
        __syncthreads();
        if (pass == 0) {
            // exclusive scan of the NH counts by wave 0 (two elements per lane)
            if (tid < 64) {
                const uint32_t a0 = 2 * tid < NH ? s_cnt[2 * tid] : 0u, a1 = 2 * tid + 1 < NH ? s_cnt[2 * tid + 1] : 0u;
                uint32_t incl = a0 + a1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_up(incl, d, 64);
                    if (tid >= d) incl += t;
                }
                const uint32_t excl = incl - (a0 + a1);
                if (2 * tid <= NH) s_cnt[2 * tid] = excl;
                if (2 * tid + 1 <= NH) s_cnt[2 * tid + 1] = excl + a0;
            }
            __syncthreads();
            for (int h = tid; h <= NH; h += GB_PREP_THREADS) ofs[size_t(rb) * ofs_stride + h] = s_cnt[h];
        }
    }
}

// what a producer keeps per record in LIST mode
struct RecL {
    const uint32_t* lst;    // the record's entry words
    const uint32_t* tile_dw;
    const pbl_rowparams* params;
    const float2* ghl;
    uint32_t ofs0, ofs1;    // lane l: ofs[l], ofs[64 + l]  (entry ranges of all half slabs)
    uint32_t cap_m1;        // last valid index of the record's entry words
    // even / odd stages keep their own registers: a stage's registers are re-requested for the stage after next right after
    // their last use, so a request has two stages to land and is never touched (copied) before it is needed
    uint32_t d[2];          // sign-plane dword
    uint32_t e[2][GB_LPF];  // entries lane, lane + 64, ... of the stage's range
    uint32_t hl[16];
};
__device__ __forceinline__ uint32_t ofs_at(const RecL& R, int h) {          // wave uniform
    return h < 64 ? __builtin_amdgcn_readlane(R.ofs0, h) : __builtin_amdgcn_readlane(R.ofs1, h - 64);
}
// The stage requests of LIST mode.  Two rules, both about `s_waitcnt vmcnt`:
//  * UNCONDITIONAL (clamped addresses instead of predicates or branches): every stage issues exactly 1 + GB_LPF loads per
//    record, so "the set requested two stages ago has landed" is `vmcnt(records per wave * (1 + GB_LPF))` -- the set requested ONE stage ago stays
//    in flight.  Words beyond the stage's range are never stored (expand_list masks by the range).
//  * issued from inline asm, waited for by wait_set(): hipcc's own bookkeeping gives up on this loop (it emitted ONE
//    `vmcnt(0)` per two stages, right after the odd stage's requests: a full memory latency exposed per iteration, and the
//    "two stages ahead" request was in effect zero to one stage ahead).  The registers are written by the asm load, named as
//    read-write operands of the wait, and only used behind it; each set is re-requested after its last use, so no copy of an
//    in-flight register is ever needed (checked in the ISA: no v_mov of d / e between request and wait).
__device__ __forceinline__ void request_stage(const RecL& R, int h, int NH, int P, int lane, uint32_t& d, uint32_t (&e)[GB_LPF]) {
    const uint32_t* pd = R.tile_dw + size_t(min(h >> 2, P - 1)) * 256 + (h & 3);      // dword (h & 3) of panel h >> 2
    asm volatile("global_load_dword %0, %1, off nt" : "=&v"(d) : "v"(pd) : "memory");
    const uint32_t lo = ofs_at(R, min(h, NH - 1));
#pragma unroll
    for (int k = 0; k < GB_LPF; ++k) {
        const uint32_t* pe = R.lst + min(lo + uint32_t(lane) + 64u * k, R.cap_m1);
        asm volatile("global_load_dword %0, %1, off" : "=&v"(e[k]) : "v"(pe) : "memory");
    }
}
static_assert(GB_LPF == 4, "wait_set spells the operands out");
__device__ __forceinline__ void wait_set(uint32_t& d0, uint32_t (&e0)[GB_LPF], uint32_t& d1, uint32_t (&e1)[GB_LPF]) {   // 2 records per wave
    asm volatile("s_waitcnt vmcnt(10)"
                 : "+v"(d0), "+v"(e0[0]), "+v"(e0[1]), "+v"(e0[2]), "+v"(e0[3]), "+v"(d1), "+v"(e1[0]), "+v"(e1[1]), "+v"(e1[2]), "+v"(e1[3])
                 :: "memory");
}
__device__ __forceinline__ void wait_set(uint32_t& d0, uint32_t (&e0)[GB_LPF]) {                                           // 1 record per wave
    asm volatile("s_waitcnt vmcnt(5)" : "+v"(d0), "+v"(e0[0]), "+v"(e0[1]), "+v"(e0[2]), "+v"(e0[3]) :: "memory");
}
__device__ __forceinline__ void store_entry(char* smem, uint32_t recaddr, uint32_t w) {
    *reinterpret_cast<uint16_t*>(smem + recaddr + (w >> 16)) = uint16_t(w & 0xFFFFu);
}
// the half slab's salient and exception words over the plane the same wave has just written
__device__ __forceinline__ void expand_list(const RecL& R, int h, const uint32_t (&e)[GB_LPF], char* smem, uint32_t recaddr, int lane) {
    const uint32_t lo = ofs_at(R, h), n = ofs_at(R, h + 1) - lo;
#pragma unroll
    for (int k = 0; k < GB_LPF; ++k)
        if (__any(uint32_t(lane) + 64u * k < n)) {             // (wave uniform: most half slabs stop after one or two)
            if (uint32_t(lane) + 64u * k < n) store_entry(smem, recaddr, e[k]);
        }
    // more than 64 * GB_LPF entries in one (record, half slab) -- rare: load and wait inside ONE asm block, invisible to the
    // compiler's vmcnt bookkeeping (a compiler-visible load in a data-dependent loop makes every wait of the stage loop vmcnt(0))
    for (uint32_t i = 64u * GB_LPF + uint32_t(lane); __any(i < n); i += 64u) {
        uint32_t w;
        const uint32_t* src = R.lst + min(lo + i, lo + n - 1u);
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(src) : "memory");
        if (i < n) store_entry(smem, recaddr, w);
    }
    asm volatile("" ::: "memory");
}
template <typename REC>
__device__ __forceinline__ void load_levels_t(REC& R, uint32_t G, uint32_t g, int lane) {
    float hi, lo;
    if (G > 1) { const float2 v = R.ghl[size_t(lane & 15) * G + g]; hi = v.x; lo = v.y; }
    else { const pbl_rowparams pr = R.params[lane & 15]; hi = pr.hi; lo = pr.lo; }
    const uint32_t hh = h16(hi), ll = h16(lo);
    const uint32_t w = (((hh - ll) & 0xFFFFu) << 16) | ll;
#pragma unroll
    for (int r = 0; r < 16; ++r) R.hl[r] = __builtin_amdgcn_readlane(w, r);
}
// the same at a group boundary INSIDE the stage loop of LIST mode: load + wait in one asm block (see expand_list)
__device__ __forceinline__ void load_levels_inloop(RecL& R, uint32_t G, uint32_t g, int lane) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v;
    const float2* src = R.ghl + size_t(lane & 15) * G + g;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");
    const uint32_t hh = h16(v.x), ll = h16(v.y);
    const uint32_t w = (((hh - ll) & 0xFFFFu) << 16) | ll;
#pragma unroll
    for (int r = 0; r < 16; ++r) R.hl[r] = __builtin_amdgcn_readlane(w, r);
}
// start-up in two phases so that the requests of BOTH records of a wave are in flight together (a wave that finishes one
// record's dependent chain before it starts the other's pays four memory round trips instead of two)
__device__ __forceinline__ uint4 init_list_a(RecL& R, const GemmArgs& a, uint32_t rb, int NH, int lane) {
    const uint8_t* blob = static_cast<const uint8_t*>(a.L.blob);
    const uint32_t* ofs = a.ofs + size_t(rb) * a.ofs_stride;
    R.ofs0 = lane <= NH ? ofs[lane] : 0u;
    R.ofs1 = lane + 64 <= NH ? ofs[lane + 64] : 0u;
    R.lst = a.lst + size_t(rb) * a.cap;
    R.cap_m1 = a.cap - 1u;
    return reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
}
__device__ __forceinline__ void init_list_b(RecL& R, const GemmArgs& a, const uint4& info, int NH, int lane) {
    const pbl_layer& L = a.L;
    const uint8_t* rec = static_cast<const uint8_t*>(L.blob) + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    R.tile_dw = reinterpret_cast<const uint32_t*>(rec + PBL_TILES_OFF(L.G)) + lane * 4;
    R.params = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
    R.ghl = reinterpret_cast<const float2*>(rec + PBL_REC_GHL_OFF);
}

// ---- consumer helpers -------------------------------------------------------------------------------------------------
struct Frag { v8h a[4], b[GB_TT]; };

__device__ __forceinline__ void load_frag(Frag& f, const char* smem, uint32_t aaddr, uint32_t baddr, bool inloop = true) {
    if ((PBL_GEMM_ABLATE & 64) && inloop) { asm volatile("" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0])); return; }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) f.a[rt] = *reinterpret_cast<const v8h*>(smem + aaddr + rt * 8192);
#pragma unroll
    for (int tt = 0; tt < GB_TT; ++tt) f.b[tt] = *reinterpret_cast<const v8h*>(smem + baddr + tt * 4096);
}

template <bool Y32, bool LIST>
__global__ __launch_bounds__((NCONS + (LIST ? NPROD_LIST : NPROD)) * GW) void pbl_gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const pbl_layer& L = a.L;
    const int K = int(L.K), M = a.M;
    const int NS = (K + PBL_SLAB_COLS - 1) / PBL_SLAB_COLS;
    const int NH = (K + GB_HS - 1) / GB_HS, NU = (K + GB_XC - 1) / GB_XC;      // half slabs, 64-column sub-steps
    // XCD-aware work order (speed only): workgroup b runs on XCD b % 8; every XCD gets a CONTIGUOUS range of the
    // token-tile-major work list, so the workgroups resident on an XCD share one 256-token slab of x in its L2.
    const uint32_t nrbk = (L.NRB + NREC - 1) / NREC, nwg = gridDim.x;
    const uint32_t xq = nwg >> 3, xr_ = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const uint32_t wg = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + xi;
    const uint32_t rowblk = wg % nrbk;
    const int tok0 = int(wg / nrbk) * GB_TOK;
#if PBL_TRACE
    uint64_t* tr = a.trace ? a.trace + (size_t(blockIdx.x) * 16 + wave) * 16 : nullptr;
    uint64_t tr_bar = 0, tr_vm = 0;
    TR_STAMP(0);
    if (tr && lane == 0) { tr[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tr[11] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
#endif

    if (wave >= NCONS) {
        // =================================== producer waves =========================================================
        const int p = wave - NCONS;
        const uint32_t gs = L.K / L.G;                            // columns per group (a multiple of 128)
#if PBL_GEMM_PPRIO
        __builtin_amdgcn_s_setprio(PBL_GEMM_PPRIO);
#endif
        if constexpr (LIST) {
            // the salient entries come ready to store from the per-call workspace (pbl_gemm_prep_kernel)
            constexpr int RPP = NREC / NPROD_LIST;                // records per producer wave
            static_assert(RPP == 1 || RPP == 2, "wait_set exists for one and two records per wave");
            RecL R[RPP];
            uint4 info[RPP];
#pragma unroll
            for (int i = 0; i < RPP; ++i) info[i] = init_list_a(R[i], a, min(rowblk * NREC + RPP * p + i, L.NRB - 1), NH, lane);
#pragma unroll
            for (int i = 0; i < RPP; ++i) init_list_b(R[i], a, info[i], NH, lane);
#pragma unroll
            for (int i = 0; i < RPP; ++i) load_levels_t(R[i], L.G, 0, lane);
            // stages 0 and 1 requested set by set: from here on exactly RPP * (1 + GB_LPF) younger loads follow a set's requests
#pragma unroll
            for (int i = 0; i < RPP; ++i) request_stage(R[i], 0, NH, int(L.P), lane, R[i].d[0], R[i].e[0]);
#pragma unroll
            for (int i = 0; i < RPP; ++i) request_stage(R[i], 1, NH, int(L.P), lane, R[i].d[1], R[i].e[1]);
            auto produce = [&](int h, auto par_tag) {             // stage h from register set PAR = h & 1
                constexpr int PAR = decltype(par_tag)::value;
                const uint32_t stage = uint32_t(PAR) * GB_AS_STAGE;
                const int kpairs = min(GB_HS, K - h * GB_HS) >> 1;
                const bool work = !(PBL_GEMM_ABLATE & 1) || h == 0;
                // this set has landed; the other one stays in flight
                {
                    TR_T0();
                    if constexpr (RPP == 2) wait_set(R[0].d[PAR], R[0].e[PAR], R[RPP - 1].d[PAR], R[RPP - 1].e[PAR]);
                    else wait_set(R[0].d[PAR], R[0].e[PAR]);
                    TR_ADD(tr_vm);
                }
#pragma unroll
                for (int i = 0; i < RPP; ++i) {
                    const uint32_t recaddr = stage + uint32_t(RPP * p + i) * 4096u;
                    if (work && !(PBL_GEMM_ABLATE & 8)) {
                        if (kpairs < GB_HS / 2) expand_sign<true>(R[i], R[i].d[PAR], smem_b, recaddr, lane, kpairs);
                        else expand_sign<false>(R[i], R[i].d[PAR], smem_b, recaddr, lane, kpairs);
                    }
                    if (work && !(PBL_GEMM_ABLATE & 16)) expand_list(R[i], h, R[i].e[PAR], smem_b, recaddr, lane);
                    request_stage(R[i], h + 2, NH, int(L.P), lane, R[i].d[PAR], R[i].e[PAR]);      // this set's next use: stage h + 2
                    if (L.G > 1 && h + 1 < NH && uint32_t((h + 1) * GB_HS) % gs == 0) load_levels_inloop(R[i], L.G, uint32_t((h + 1) * GB_HS) / gs, lane);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the stage is in LDS
                if (h == 0) TR_STAMP(1);
                {
                    TR_T0();
                    if (!(PBL_GEMM_ABLATE & 128) || h == 0) __builtin_amdgcn_s_barrier();
                    TR_ADD(tr_bar);
                }
                asm volatile("" ::: "memory");
            };
            for (int h = 0; h < NH; h += 2) {
                produce(h, std::integral_constant<int, 0>{});
                if (h + 1 < NH) produce(h + 1, std::integral_constant<int, 1>{});
            }
            TR_STAMP(2);
            TR_STAMP(3);
#if PBL_TRACE
            if (tr && lane == 0) { tr[8] = tr_bar; tr[9] = tr_vm; }
#endif
            return;
        }
        Rec R[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) init_rec(R[i], L, min(rowblk * NREC + 2 * p + i, L.NRB - 1), lane);   // (a record beyond the layer mirrors the last one; its rows are never stored)
        // stage 0, then one stage ahead of the consumers; barrier b (b = 0 .. NH-1) closes stage b
        for (int h = 0; h < NH; ++h) {
            const uint32_t stage = uint32_t(h & 1) * GB_AS_STAGE;
            const int kpairs = min(GB_HS, K - h * GB_HS) >> 1;
            const bool work = !(PBL_GEMM_ABLATE & 1) || h == 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t recaddr = stage + uint32_t(2 * p + i) * 4096u;
                if (work && !(PBL_GEMM_ABLATE & 8)) {
                    if (kpairs < GB_HS / 2) expand_sign<true>(R[i], R[i].dcur, smem_b, recaddr, lane, kpairs);
                    else expand_sign<false>(R[i], R[i].dcur, smem_b, recaddr, lane, kpairs);
                }
                if (work && !(PBL_GEMM_ABLATE & 16)) expand_sal(R[i], h, smem_b, recaddr, lane);
            }
            // requests for stage h + 2 (the sign-plane dword) and, when stage h + 1 opens a slab, for the slab after that one
            // (1.5 slabs ahead); what stage h + 1 needs was requested an iteration ago
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (PBL_GEMM_ABLATE & 32) break;
                R[i].dcur = R[i].dnxt;
                R[i].dnxt = load_dw(R[i], h + 2, int(L.P));
                if (h & 1) {
                    adopt_slab(R[i], lane & 3);
                    request_slab(R[i], R[i].sdn, ((h + 1) >> 1) + 1, NS, lane & 3);
                }
                if (L.G > 1 && h + 1 < NH && uint32_t((h + 1) * GB_HS) % gs == 0) load_levels(R[i], L.G, uint32_t((h + 1) * GB_HS) / gs, lane);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the stage is in LDS
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        return;
    }

    // =================================== consumer waves =============================================================
    const int c = wave;
    const int i32 = lane & 31, g = lane >> 5;
    const uint32_t xring = GB_X_OFF + uint32_t(c) * GB_XRING_BYTES;
    // x through a buffer descriptor that starts at this workgroup's first token: tokens >= M read zeros
    const char* xbase = reinterpret_cast<const char*>(a.x) + size_t(tok0) * size_t(K) * 2;
    const size_t xrem = size_t(min(M - tok0, GB_TOK)) * size_t(K) * 2;          // <= 256 * 32767 * 2 < 2^24
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase), 0, int(xrem), 0x00020000);
    // DMA piece q (1 KiB) of a slot = tokens 8q .. 8q+7 x 128 B; lane l lands on unit l & 7 of token 8q + (l >> 3), which holds
    // the LOGICAL unit (l & 7) ^ ((token >> 1) & 7)
    uint32_t xvoff[GB_PQ];
#pragma unroll
    for (int q = 0; q < GB_PQ; ++q) {
        const uint32_t tl = uint32_t(8 * q + (lane >> 3));                           // token within the wave's own
        xvoff[q] = (uint32_t(GB_TPC * c) + tl) * uint32_t(K) * 2u + ((uint32_t(lane & 7) ^ ((tl >> 1) & 7)) << 4);
    }
    // K % 64 != 0: the units of the LAST sub-step that lie beyond K would hold the next token row; their source offset is
    // pushed out of the descriptor's range instead, so they read zeros (and the producers zero the weights there too).
    // Piece q, lane l holds logical unit (l & 7) ^ (((l >> 4) | 4 (q & 1))): two lane masks, for even and odd q.
    const uint32_t ktail_units = uint32_t(K & (GB_XC - 1)) >> 3;                 // valid units of the last sub-step (0: no tail)
    uint32_t xbad[2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
        xbad[o] = (ktail_units && ((uint32_t(lane & 7) ^ (uint32_t(lane >> 4) | uint32_t(4 * o))) >= ktail_units)) ? 0x40000000u : 0u;
    auto issue_x = [&](int u, int q) {       // piece q of sub-step u into ring slot u % 3
        const uint32_t dst = xring + uint32_t(u % GB_XSLOTS) * GB_XSLOT_BYTES + uint32_t(q) * 1024u;
        const uint32_t vo = xvoff[q] + uint32_t(u) * (GB_XC * 2) + (u == NU - 1 ? xbad[q & 1] : 0u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(smem_b + dst), 16, int(vo), 0, 0, 0);
    };
    // fragment addresses: A unit (2 ks8 + g) ^ (i32 & 15) of row i32 (+ 32 rt), ks8 = 4 (u & 1) + k: the odd sub-step's units
    // are the even one's with bit 3 flipped (byte 128); x unit (2 k + g) ^ ((i32 >> 1) & 7) of token i32 (+ 32 tt)
    uint32_t aq[4], bq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) aq[k] = uint32_t(i32) * 256u + ((uint32_t(2 * k + g) ^ uint32_t(i32 & 15)) << 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) bq[k] = xring + uint32_t(i32) * 128u + ((uint32_t(2 * k + g) ^ uint32_t((i32 >> 1) & 7)) << 4);

    v16f acc[4][GB_TT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int tt = 0; tt < GB_TT; ++tt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][tt][e] = 0.f;

    // prologue: the first three sub-steps of x on their way, stage 0 of A behind barrier 0
#pragma unroll
    for (int u = 0; u < GB_XSLOTS; ++u)
        if (u < NU)
#pragma unroll
            for (int q = 0; q < GB_PQ; ++q) issue_x(u, q);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // slot 0 has landed when at most the pieces of sub-steps 1, 2 are outstanding
    if (NU >= 3) { if (GB_PQ == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else if (NU == 2) { if (GB_PQ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_STAMP(1);
#if PBL_GEMM_PRIO
    __builtin_amdgcn_s_setprio(PBL_GEMM_PRIO);
#endif
#if PBL_GEMM_CPRIO_YOUNG
    if (c >= NCONS / 2) __builtin_amdgcn_s_setprio(PBL_GEMM_CPRIO_YOUNG);
#endif
    auto alt_prio = [&](int k) {
#if PBL_GEMM_ALT_PRIO
        if (((k & 1) != 0) == (c >= NCONS / 2)) __builtin_amdgcn_s_setprio(PBL_GEMM_ALT_PRIO);
        else __builtin_amdgcn_s_setprio(0);
#endif
    };
    Frag f0, f1;
    load_frag(f0, smem_b, aq[0], bq[0], false);      // (stage 0, first half: no offsets)
    if (PBL_GEMM_ABLATE & 64) load_frag(f1, smem_b, aq[1], bq[1], false);

    // One 64-column sub-step = 4 k-steps of 16 columns; fragments of k-step kk+1 are read while k-step kk multiplies.
    auto mma = [&](const Frag& f) {
        if (!(PBL_GEMM_ABLATE & 4)) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int tt = 0; tt < GB_TT; ++tt)
                    acc[rt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[rt], f.b[tt], acc[rt][tt], 0, 0, 0);
        }
    };
    for (int u = 0; u < NU; ++u) {
        const uint32_t abuf = uint32_t((u >> 1) & 1) * GB_AS_STAGE, ahalf = uint32_t(u & 1) * 128u;
        const uint32_t xslot = uint32_t(u % GB_XSLOTS) * GB_XSLOT_BYTES;
        const bool last = u + 1 >= NU;
        // pieces of sub-step u+2's x go out GB_PQ / 4 per k-step: its slot, (u+2) % 3 == (u-1) % 3, was last read in sub-step u-1
        const bool stage_x = !(PBL_GEMM_ABLATE & 2) && u >= 1 && u + 2 < NU;
        // k-step 0
        alt_prio(0);
        load_frag(f1, smem_b, (aq[1] ^ ahalf) + abuf, bq[1] + xslot);
        if (stage_x) { issue_x(u + 2, 0 * (GB_PQ / 4)); if (GB_PQ == 8) issue_x(u + 2, 0 * 2 + 1); }
        mma(f0);
        // k-step 1
        alt_prio(1);
        load_frag(f0, smem_b, (aq[2] ^ ahalf) + abuf, bq[2] + xslot);
        if (stage_x) { issue_x(u + 2, 1 * (GB_PQ / 4)); if (GB_PQ == 8) issue_x(u + 2, 1 * 2 + 1); }
        mma(f1);
        // k-step 2
        alt_prio(0);
        load_frag(f1, smem_b, (aq[3] ^ ahalf) + abuf, bq[3] + xslot);
        if (stage_x) { issue_x(u + 2, 2 * (GB_PQ / 4)); if (GB_PQ == 8) issue_x(u + 2, 2 * 2 + 1); }
        mma(f0);
        // k-step 3: every read of this sub-step's x slot -- and, in an odd sub-step, of the A stage -- has been issued
        alt_prio(1);
        if (stage_x) { issue_x(u + 2, 3 * (GB_PQ / 4)); if (GB_PQ == 8) issue_x(u + 2, 3 * 2 + 1); }
        if (!last) {
            // x of sub-step u+1: its pieces were issued during sub-step u-1 (or in the prologue); younger: sub-step u+2's
            {
                TR_T0();
                if (u + 2 < NU) { if (GB_PQ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TR_ADD(tr_vm);
            }
            if ((u & 1) && !(PBL_GEMM_ABLATE & 128)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of the stage have returned
                TR_T0();
                __builtin_amdgcn_s_barrier();                            // ... the next stage is complete
                TR_ADD(tr_bar);
                asm volatile("" ::: "memory");
            }
            const uint32_t nabuf = uint32_t(((u + 1) >> 1) & 1) * GB_AS_STAGE, nahalf = uint32_t((u + 1) & 1) * 128u;
            load_frag(f0, smem_b, (aq[0] ^ nahalf) + nabuf, bq[0] + uint32_t((u + 1) % GB_XSLOTS) * GB_XSLOT_BYTES);
        }
        mma(f1);
    }
#if PBL_GEMM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif

    // ---- epilogue, per consumer wave: accumulators (+ bias) -> Ys[tokens][128 rows] in the wave's own ring -> 16-byte stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    TR_STAMP(2);
#if PBL_TRACE
    if (tr && lane == 0) { tr[8] = tr_bar; tr[9] = tr_vm; }
#endif
    typedef typename std::conditional<Y32, float, _Float16>::type yt;
    constexpr uint32_t YSTR = Y32 ? 528u : 272u;              // bytes per token row: 128 rows + 16 B (every 16-byte read-back stays aligned)
    constexpr int EPT = (32u * YSTR <= GB_XRING_BYTES) ? 32 : 16;     // tokens per pass: what the ring holds (fp32 result of 8 consumers: 16)
    const uint32_t row0 = rowblk * GB_ROWS;
    const bool vec = (L.N & (Y32 ? 3 : 7)) == 0 && row0 + GB_ROWS <= L.N;     // whole 16-byte units, all rows exist
    if (PBL_GEMM_ABLATE & 512) {          // (timing probe: no epilogue at all; the accumulators stay live)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int tt = 0; tt < GB_TT; ++tt) asm volatile("" :: "v"(acc[rt][tt]));
        return;
    }
#pragma unroll
    for (int tt = 0; tt < GB_TT; ++tt) {
#pragma unroll
        for (int ph = 0; ph < 32 / EPT; ++ph) {
            if (EPT == 32 || (i32 >> 4) == ph) {              // this lane's token (i32) belongs to the pass
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int rloc = rt * 32 + 8 * q4 + 4 * g;        // 4 consecutive rows held by this lane: D row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                        yt h[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t row = row0 + uint32_t(rloc + r);
                            const float b = (L.bias && row < L.N) ? L.bias[row] : 0.f;
                            h[r] = yt(acc[rt][tt][4 * q4 + r] + b);
                        }
                        char* dst = smem_b + xring + uint32_t(i32 & (EPT - 1)) * YSTR + uint32_t(rloc) * sizeof(yt);
                        if (Y32) *reinterpret_cast<v4f*>(dst) = v4f{float(h[0]), float(h[1]), float(h[2]), float(h[3])};
                        else {
                            uint2 pk;
                            pk.x = uint32_t(__builtin_bit_cast(uint16_t, _Float16(h[0]))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(h[1]))) << 16);
                            pk.y = uint32_t(__builtin_bit_cast(uint16_t, _Float16(h[2]))) | (uint32_t(__builtin_bit_cast(uint16_t, _Float16(h[3]))) << 16);
                            *reinterpret_cast<uint2*>(dst) = pk;
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int UPR = GB_ROWS * int(sizeof(yt)) / 16;      // 16-byte units per token row: 16 (fp16) / 32 (fp32)
            for (int idx = lane; idx < EPT * UPR; idx += GW) {
                const int t = idx / UPR, un = idx % UPR, tok = tok0 + GB_TPC * c + 32 * tt + EPT * ph + t;
                if (tok >= M) continue;
                constexpr int EPU = 16 / int(sizeof(yt));            // elements per unit
                yt* dstg = static_cast<yt*>(a.y) + size_t(tok) * L.N + row0 + un * EPU;
                const yt* src = reinterpret_cast<const yt*>(smem_b + xring + uint32_t(t) * YSTR) + un * EPU;
                if (PBL_GEMM_ABLATE & 256) { if (tok == 0x7FFFFFFF) *reinterpret_cast<u32x4*>(dstg) = *reinterpret_cast<const u32x4*>(src); }    // (timing probe: no result stores)
                else if (vec) *reinterpret_cast<u32x4*>(dstg) = *reinterpret_cast<const u32x4*>(src);
                else
                    for (int e = 0; e < EPU; ++e)
                        if (row0 + un * EPU + e < L.N) dstg[e] = src[e];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done before the next pass overwrites the buffer
        }
    }
#if PBL_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the result stores have left the wave)
    TR_STAMP(3);
#endif
}

}  // namespace

namespace {
size_t align16(size_t v) { return (v + 15) & ~size_t(15); }
// entry words per record; never 0: the producers' unconditional requests clamp their index to cap - 1, so a layer without any
// salient entry (fully binarized) still owns one 16-byte line per record
uint32_t list_cap(const pbl_layer* l) { const uint32_t c = (16u * l->max_nch + l->max_nexc + 3u) & ~3u; return c ? c : 4u; }
int check_layer(const pbl_layer* layer, const void* x, const void* y, int M) {
    if (!layer || !layer->blob || !x || !y || M < 1) return PBL_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(layer->blob) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
        return PBL_ERR_MISALIGNED;
    if ((layer->K & 7) || !(layer->flags & PBL_FLAG_SLABS) || !(layer->flags & PBL_FLAG_TAIL_REPEAT)) return PBL_ERR_UNSUPPORTED;
    if (layer->G < 1 || (layer->G > 1 && (layer->K % layer->G || (layer->K / layer->G) % GB_HS))) return PBL_ERR_UNSUPPORTED;
    return PBL_OK;
}
}  // namespace

// Bytes of the salient list of a layer (LIST mode): per-(record, half slab) entry ranges + one 4-byte word per salient entry
// and exception.  It depends on the layer only, never on x: pbl_gemm_prepare builds it, pbl_gemm_f16_prepared consumes it any
// number of times.  0: the layer is too wide for the range registers (more than 127 half slabs) -- no list, the GEMM kernel
// decodes in place.
extern "C" size_t pbl_gemm_list_bytes(const pbl_layer* layer) {
    if (!layer) return 0;
    const uint32_t NH = (layer->K + GB_HS - 1) / GB_HS;
    if (NH > GB_LIST_MAX_NH) return 0;
    const size_t ofs_stride = (NH + 1 + 3) & ~size_t(3);
    return align16(size_t(layer->NRB) * ofs_stride * 4) + size_t(layer->NRB) * list_cap(layer) * 4;
}

// Bytes of transient device workspace pbl_gemm_f16_ws wants for M rows of x: the list above when the call spans more than one
// 256-token tile (every tile would decode the same entries again), else 0 (decode inside the GEMM kernel).
extern "C" size_t pbl_gemm_workspace_bytes(const pbl_layer* layer, int M) {
    if (!layer || M <= GB_TOK) return 0;
    return pbl_gemm_list_bytes(layer);
}

namespace {
bool list_fits(const pbl_layer* layer, const void* workspace, size_t workspace_bytes) {
    const size_t want = pbl_gemm_list_bytes(layer);
    return workspace && want && workspace_bytes >= want && !(reinterpret_cast<uintptr_t>(workspace) & 15);
}
void list_views(const pbl_layer* layer, void* workspace, GemmArgs& a) {
    const uint32_t NH = (layer->K + GB_HS - 1) / GB_HS;
    a.ofs_stride = (NH + 1 + 3) & ~3u;
    a.cap = list_cap(layer);
    a.ofs = static_cast<uint32_t*>(workspace);
    a.lst = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + align16(size_t(layer->NRB) * a.ofs_stride * 4));
}
int launch_gemm(GemmArgs& a, bool list, hipStream_t s) {
#if PBL_TRACE
    a.trace = g_gemm_trace;
#endif
    const void* k = list ? (a.y_f32 ? reinterpret_cast<const void*>(pbl_gemm_kernel<true, true>) : reinterpret_cast<const void*>(pbl_gemm_kernel<false, true>))
                         : (a.y_f32 ? reinterpret_cast<const void*>(pbl_gemm_kernel<true, false>) : reinterpret_cast<const void*>(pbl_gemm_kernel<false, false>));
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(GB_LDS)) != hipSuccess) return PBL_ERR_LAUNCH;
    void* argv[] = {&a};
    const dim3 grid(((a.L.NRB + NREC - 1) / NREC) * uint32_t((a.M + GB_TOK - 1) / GB_TOK));
    return hipLaunchKernel(k, grid, dim3((NCONS + (list ? NPROD_LIST : NPROD)) * GW), argv, GB_LDS, s) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}
}  // namespace

// Build the layer's salient list into `workspace` (>= pbl_gemm_list_bytes(layer), 16-byte aligned): one small kernel, no x.
// The list stays valid as long as the blob is unchanged; a caller may keep it per layer (4 B per salient entry) or build it
// for the NEXT layer on a second stream while this layer's GEMM runs.
extern "C" int pbl_gemm_prepare(const pbl_layer* layer, void* workspace, size_t workspace_bytes, void* stream) {
    if (!layer || !layer->blob) return PBL_ERR_INVALID_ARG;
    if ((layer->K & 7) || !(layer->flags & PBL_FLAG_SLABS) || !(layer->flags & PBL_FLAG_TAIL_REPEAT)) return PBL_ERR_UNSUPPORTED;
    if (!pbl_gemm_list_bytes(layer)) return PBL_ERR_UNSUPPORTED;
    if (!list_fits(layer, workspace, workspace_bytes)) return workspace && workspace_bytes >= pbl_gemm_list_bytes(layer) ? PBL_ERR_MISALIGNED : PBL_ERR_INVALID_ARG;
    GemmArgs a;
    list_views(layer, workspace, a);
    pbl_layer lcopy = *layer;
    uint32_t* ofs = const_cast<uint32_t*>(a.ofs);
    uint32_t* lst = const_cast<uint32_t*>(a.lst);
    void* pargv[] = {&lcopy, &ofs, &lst, &a.ofs_stride, &a.cap};
    const size_t plds = size_t(2 * (GB_LIST_MAX_NH + 1)) * 4 + 32 * 4 + 16 * sizeof(pbl_rowinfo) + ((size_t(layer->max_nch) + 15) & ~size_t(15));
    return hipLaunchKernel(reinterpret_cast<const void*>(pbl_gemm_prep_kernel), dim3(layer->NRB), dim3(GB_PREP_THREADS), pargv, plds,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

// The GEMM over a list pbl_gemm_prepare built for THIS layer (any M >= 1).  Bit-identical to pbl_gemm_f16_ws / _ex.
extern "C" int pbl_gemm_f16_prepared(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, const void* workspace,
                                     size_t workspace_bytes, void* stream) {
    const int st = check_layer(layer, x, y, M);
    if (st != PBL_OK) return st;
    if (!list_fits(layer, workspace, workspace_bytes)) return PBL_ERR_INVALID_ARG;
    GemmArgs a;
    a.L = *layer; a.x = static_cast<const _Float16*>(x); a.y = y; a.M = M; a.y_f32 = y_f32;
    list_views(layer, const_cast<void*>(workspace), a);
    return launch_gemm(a, true, static_cast<hipStream_t>(stream));
}

extern "C" int pbl_gemm_f16_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* workspace, size_t workspace_bytes,
                               void* stream) {
    const int st = check_layer(layer, x, y, M);
    if (st != PBL_OK) return st;
    if (pbl_gemm_workspace_bytes(layer, M) && list_fits(layer, workspace, workspace_bytes)) {
        const int sp = pbl_gemm_prepare(layer, workspace, workspace_bytes, stream);
        if (sp != PBL_OK) return sp;
        return pbl_gemm_f16_prepared(layer, x, y, M, y_f32, workspace, workspace_bytes, stream);
    }
    GemmArgs a;
    a.L = *layer; a.x = static_cast<const _Float16*>(x); a.y = y; a.M = M; a.y_f32 = y_f32;
    a.ofs = nullptr; a.lst = nullptr; a.ofs_stride = 0; a.cap = 0;
    return launch_gemm(a, false, static_cast<hipStream_t>(stream));
}

extern "C" int pbl_gemm_f16_ex(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* stream) {
    return pbl_gemm_f16_ws(layer, x, y, M, y_f32, nullptr, 0, stream);
}

extern "C" int pbl_gemm_f16(const pbl_layer* layer, const void* x, void* y, int M, void* stream) {
    return pbl_gemm_f16_ws(layer, x, y, M, 0, nullptr, 0, stream);
}
