// pbl_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of libpbl.so.
//
// One kernel family: the partially-binarized GEMV
//     y[m, r] = sum_j v_rj * x[m, j] + bias_r            (m <= 4 tokens per pass)
// over the PBL1 packed format (include/pbl.h).  It replaces the reference's
//     w_sim = where(mask, W*outlier_scale, sign(W)*binary_scale); F.linear(x, w_sim, b)
// (quant/outlier_quantizer.py:83-106) and the stock nn.Linear forward over GPTQ-PB
// fake-quant weights (gptq_pb/gptq.py:180-184).
//
// Work decomposition: ONE WAVEFRONT (64 lanes) owns ONE RECORD = 16 output rows.
//   phase 0  the workgroup stages x (fp16) into LDS, zero padded
//   phase 1  sign plane: per 512-column panel one coalesced 1 KiB global_load_dwordx4
//            per wave; lane l holds x for ITS columns in 4 VGPRs (half2), unpacks a
//            pair of weights to a two-valued fp16 constant with ONE v_and(_or)_b32
//            ("bit classes") and accumulates with v_dot2_f32_f16
//   phase 2  salient list: lane-per-chunk (16 delta-coded columns + 16 uint8 codes,
//            coalesced 16 B/lane loads), x gathered from LDS
//   phase 3  transpose-reduce the 16 row accumulators across the wave, combine
//            with the per-chunk partials in fixed order (deterministic), store y.
// HBM-bound by design: every weight byte is read exactly once, x comes from L2/LDS.
// No inter-workgroup communication, so no XCD-placement dependence.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"
#include "pbl_p2p_layout.h"

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define PBL_WAVE 64
// Ablation builds for performance analysis only (tools/ablate.sh): 1 = memory only
// (loads kept live, math skipped), 2 = compute only (weight-stream loads skipped).
// TIMING-ONLY probes for tools/build_variant.sh (results are wrong; 0 in every shipped build): bit 0 chunk partials at half size
// (one more workgroup per CU), 1 no per-word sum of x, 2 classes 8 / 9 unpacked by a plain AND, 3 tile loads without nt,
// 4 salient loads without nt
#ifndef PBL_PROBE
#define PBL_PROBE 0
#endif
#ifndef PBL_ABLATE
#define PBL_ABLATE 0
#endif
// pbl_unpack_dev: stream the dense rows out with non-temporal stores (A/B builds; the library GEMM that follows reads the
// buffer right away, so bypassing the caches is not obviously right)
#ifndef PBL_UNPACK_NT
#define PBL_UNPACK_NT 0
#endif

namespace {

typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ T load_tile(const T* p) { return (PBL_PROBE & 8) ? *p : __builtin_nontemporal_load(p); }
template <typename T> __device__ __forceinline__ T load_sal(const T* p) { return (PBL_PROBE & 16) ? *p : __builtin_nontemporal_load(p); }

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}

// fl16(fl32(a*b)): the checkpoint's value of a salient weight is DOUBLE rounded (fp32 product,
// then `.to(fp16)`, gptq_pb/gptq.py:182).  hipcc would fuse mul + cvt into one single-rounding
// v_fma_mixlo_f16 (1 fp16 ulp off on ~1e-5 of the values); the empty asm keeps the fp32 product.
__device__ __forceinline__ _Float16 round_f16_twice(float prod) {
    asm volatile("" : "+v"(prod));
    return _Float16(prod);
}

// Bit classes 8..15 of an fp16 half-word: pair mask M and OR-constant C such that
// (w & M) | C is an fp16x2 holding {lo_c, lo_c + d_c} per element.  The signed
// sum over a lane's columns is  D = A_c * acc - B_c * Xl  (Xl = plain sum of x).
//   c=8 : {1,1.25}  c=9 : {1,1.5}  c=10..14 : {0, 2^-14, 2^-13, 2^-11, 2^-7, 2}  c=15 : {1,-1}
template <int CI> struct BitClass;
#define PBL_CLASS(ci, m, c) \
    template <> struct BitClass<ci> { static constexpr uint32_t M = m, C = c; };
PBL_CLASS(0, 0x01000100u, 0x3C003C00u)
PBL_CLASS(1, 0x02000200u, 0x3C003C00u)
PBL_CLASS(2, 0x04000400u, 0u)
PBL_CLASS(3, 0x08000800u, 0u)
PBL_CLASS(4, 0x10001000u, 0u)
PBL_CLASS(5, 0x20002000u, 0u)
PBL_CLASS(6, 0x40004000u, 0u)
PBL_CLASS(7, 0x80008000u, 0x3C003C00u)
#undef PBL_CLASS

__device__ __forceinline__ void class_consts(int ci, float& A, float& B) {
    const float a[8] = {8.f, 4.f, 32768.f, 16384.f, 4096.f, 256.f, 1.f, -1.f};
    const float b[8] = {9.f, 5.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f};
    A = a[0]; B = b[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { A = ci == k ? a[k] : A; B = ci == k ? b[k] : B; }
}

// (w & M) | C in one VALU op.  hipcc splits it into v_and + v_or because VOP3 takes no
// literal on gfx9; with M in an SGPR and C in a VGPR it is a single v_and_or_b32.
template <int CI>
__device__ __forceinline__ uint32_t unpack_pair(uint32_t w, uint32_t c_one, float order_after) {
    if constexpr (BitClass<CI>::C == 0u || ((PBL_PROBE & 4) && CI < 2)) {
        return w & BitClass<CI>::M;
    } else {
        // `order_after` (an accumulator the previous class step wrote) is an unused input: it pins this instruction behind that
        // step.  Without it hipcc schedules all 24 v_and_or of a tile first (to re-request the tile registers early), holds 24
        // temporaries and the kernel drops from 8 to 6 waves per SIMD.
        uint32_t t;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(t) : "v"(w), "s"(BitClass<CI>::M), "v"(c_one), "v"(order_after));
        return t;
    }
}

template <int MB, int CI>
__device__ __forceinline__ void class_step(uint32_t w, uint32_t ws, uint32_t c_one, const uint32_t (&xr)[MB],
                                           float (&acc)[MB][16]) {
    const float dep = acc[0][(CI + 7) & 7];
    const uint32_t t0 = unpack_pair<CI>(w, c_one, dep);
    const uint32_t t1 = unpack_pair<CI>(ws, c_one, dep);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        acc[m][CI] = dot2(t0, xr[m], acc[m][CI]);
        acc[m][8 + CI] = dot2(t1, xr[m], acc[m][8 + CI]);
    }
}

template <int MB>
__device__ __forceinline__ void word_step(uint32_t w, uint32_t c_one, const uint32_t (&xr)[MB],
                                          float (&acc)[MB][16], float (&xl)[MB]) {
    const uint32_t ws = w << 8;
    if (!(PBL_PROBE & 2))
#pragma unroll
        for (int m = 0; m < MB; ++m) xl[m] = dot2(c_one, xr[m], xl[m]);
    class_step<MB, 0>(w, ws, c_one, xr, acc);
    class_step<MB, 1>(w, ws, c_one, xr, acc);
    class_step<MB, 2>(w, ws, c_one, xr, acc);
    class_step<MB, 3>(w, ws, c_one, xr, acc);
    class_step<MB, 4>(w, ws, c_one, xr, acc);
    class_step<MB, 5>(w, ws, c_one, xr, acc);
    class_step<MB, 6>(w, ws, c_one, xr, acc);
    class_step<MB, 7>(w, ws, c_one, xr, acc);
}

// DPP / permlane helpers (wave64, gfx950).
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_ROR8 = 0x128,
              DPP_ROW_MIRROR = 0x140;

// v_permlane{32,16}_swap exchange halves / odd-even rows of TWO registers in place.  The
// __builtin forms mis-pair their two results under hipcc 7.2 here (the second result is
// read from the first register), so the swap is issued from inline asm; "s_nop 1" covers
// the VALU-write -> permlane-read hazard, which the compiler does not pad inside asm.
// sum of a over the two 32-lane halves -> lanes 0..31 get a's total, lanes 32..63 b's
__device__ __forceinline__ float fold32(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// same across the odd/even 16-lane rows: even rows get a's total, odd rows get b's
__device__ __forceinline__ float fold16(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// 16 per-lane accumulators -> lane l holds the wave total of row
//   rho(l) = 8*bit5(l) + 4*bit4(l) + 2*bit3(l) + bit2(l)   (4 lanes per row).
__device__ __forceinline__ float transpose_reduce16(const float (&a)[16], int lane) {
    float v[8], u[4], t[2];
    const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fold32(a[j], a[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = fold16(v[j], v[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float send = b3 ? u[j] : u[j + 2], keep = b3 ? u[j + 2] : u[j];
        t[j] = keep + dpp<DPP_ROW_ROR8>(send);
    }
    const float send = b2 ? t[0] : t[1], keep = b2 ? t[1] : t[0];
    float s = keep + dpp<DPP_HALF_MIRROR>(send);  // partner 7-i: opposite bit2, bijective on bits 0..1
    s += dpp<DPP_QUAD_XOR1>(s);
    s += dpp<DPP_QUAD_XOR2>(s);
    return s;
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp<DPP_QUAD_XOR1>(v);
    v += dpp<DPP_QUAD_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    v += dpp<DPP_ROW_MIRROR>(v);
    v = fold16(v, v);
    return fold32(v, v);
}

// sum over the 4 lanes that share a row (lane bits 0..1)
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp<DPP_QUAD_XOR1>(v);
    return v + dpp<DPP_QUAD_XOR2>(v);
}

struct GemvArgs {
    pbl_layer layer;      // single-layer launch: by value
    const _Float16* x;    // [M, K]
    void* y;              // [M, N] fp16 or fp32
    // grouped launch: device arrays
    const pbl_layer* layers;
    const void* const* xs;
    void* const* ys;
    int M;                // tokens in this pass (== MB)
    int y_f32;
    int grouped;
    // fused launch (pbl_gemv_f16_fused): every layer reads the SAME x and writes its column range of ONE output
    // matrix y[M, ldy]; x_shared / y_shared override xs[] / ys[] when set
    const _Float16* x_shared;
    void* y_shared;
    const uint64_t* y_off;   // device: element offset of layer l's columns in a row of y
    uint32_t ldy;            // row stride of y in elements (0: the layer's own N)
    int Lc;                  // layers of a grouped launch (host side only)
    int tok0;                // column-group kernel in a grouped launch: first token of this pass (it takes 2 tokens per pass)
    // fused launch of at most PBL_FUSED_INLINE_MAX layers: the descriptors and output offsets travel IN the kernel arguments
    // (scalar loads from the argument segment) instead of a device table -- one dependent memory round trip less per workgroup
    int n_inl;
    pbl_layer inl[PBL_FUSED_INLINE_MAX];
    uint64_t inl_off[PBL_FUSED_INLINE_MAX];
    // fused peer push (pbl_linear_f16_push; K-split tensor parallelism): the row owners write their fp32 partials straight into
    // slot [set][rank] of EVERY rank's communication buffer (csrc/pbl_p2p_layout.h) instead of y, and every record counts itself
    // at the peers once its rows are visible system wide; pbl_p2p_reduce_f32_dev sums the slots.  push_world == 0: off.
    int push_world, push_rank;
    size_t push_cap;
    uint8_t* push_peer[PBL_P2P_MAX_WORLD];
    // bf16 activations IN the kernel (round 5; pbl_linear_bf16 / pbl_gemv_bf16_fused_host): x is bf16 -- the staging phase finds every
    // token's largest magnitude, scales the row by a power of two into fp16's range (exact) while it copies it into LDS, the epilogue
    // multiplies the result back and writes bf16 (fp32 when y_f32).  A token holding inf / NaN runs as its indicator row with scale
    // +inf.  Exactly the arithmetic of pbl_act_bf16_prepare + the fp16 kernel + pbl_act_finish (csrc/pbl_act.hip), in ONE launch.
    int x_bf16;
};

// Gather 8 fp16 values from LDS byte addresses a[0..7] into 4 packed half2 registers
// (entry 2p in the low half, 2p+1 in the high half).  All loads are in flight and waited
// for once inside the statement (guide 5.7 form (i)); hipcc's own schedule waits per load.
// d16/d16_hi loads cannot build the pair: with SRAM-ECC (always on here) they ZERO the
// other half of the destination instead of preserving it, so the halves are merged with
// one v_lshl_or_b32 per pair.
__device__ __forceinline__ void gather8(const uint32_t (&a)[8], uint32_t (&x)[4]) {
    uint32_t t[8];
    asm volatile(
        "ds_read_u16 %0, %8\n\tds_read_u16 %1, %9\n\tds_read_u16 %2, %10\n\tds_read_u16 %3, %11\n\t"
        "ds_read_u16 %4, %12\n\tds_read_u16 %5, %13\n\tds_read_u16 %6, %14\n\tds_read_u16 %7, %15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
        : "memory");
#pragma unroll
    for (int p = 0; p < 4; ++p) x[p] = (t[2 * p + 1] << 16) | t[2 * p];
}

// One salient chunk (lane-private): 16 delta-coded columns + 16 uint8 codes, processed as
// two halves of 4 pairs (keeps the live register set small -> more waves per SIMD).
// Codes become fp16 (1024+q) with ONE v_perm_b32 per pair (0x64 exponent byte), so
//   Qb += dot2((1024+q0, 1024+q1), (x0, x1))   and   S += dot2((1,1), (x0,x1));
// the true code sum is Qb - 1024*S.  Deltas are stored pre-doubled (byte steps into the
// fp16 x tile), so one SDWA add per entry yields the LDS address.
//
// SF (PBL_FLAG_SAL_F16 layers, fp16 checkpoints): the salient weight is fl16(ss*(q-sz)); it
// is rebuilt per entry (v_cvt_f32_ubyte, v_sub, v_mul) and the pair rounded with ONE
// v_cvt_pk_f16_f32, so Q accumulates the exact checkpoint value times x.
template <int MB, bool PRED, bool SF>
__device__ __forceinline__ void chunk_accumulate(uint32_t xbase, uint32_t tok_stride_bytes, uint32_t col0,
                                                 const u32x4& d4, const u32x4& q4, int cnt, uint32_t zaddr,
                                                 uint32_t c_one, float ss, float sz, float (&Q)[MB], float (&S)[MB]) {
    uint32_t run = xbase + 2u * col0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = 8 * h + k;
            run += (d4[e >> 2] >> (8 * (e & 3))) & 0xFFu;
            a[k] = PRED ? (e < cnt ? run : zaddr) : run;
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            uint32_t x[4];
            gather8(a, x);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t qp;
                if constexpr (SF) {
                    const uint32_t qw = q4[2 * h + (p >> 1)] >> ((p & 1) * 16);
                    h2 w;
                    w.x = round_f16_twice(ss * (float(qw & 0xFFu) - sz));
                    w.y = round_f16_twice(ss * (float((qw >> 8) & 0xFFu) - sz));
                    qp = __builtin_bit_cast(uint32_t, w);
                } else {
                    qp = __builtin_amdgcn_perm(q4[2 * h + (p >> 1)], 0x64646464u, (p & 1) ? 0x00070006u : 0x00050004u);
                }
                Q[m] = dot2(qp, x[p], Q[m]);
                S[m] = dot2(c_one, x[p], S[m]);
            }
            if (m + 1 < MB) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += tok_stride_bytes;
            }
        }
    }
}

#ifndef PBL_TILE_RING
#define PBL_TILE_RING 2          // 2: rotating tile registers (63 VGPRs, 8 waves/SIMD; hipcc waits vmcnt(0) at the rotation, so
                                 // the lookahead is one panel).  3: static three-set ring, a true two-panel lookahead at 74 VGPRs
                                 // = 6 waves/SIMD.  Measured equal on the driver command (round 3, profiles/r03_gemv_ring.md):
                                 // 0.669-0.688 of the HBM roofline for 3 vs 0.663-0.680 for 2 -- kept selectable, default unchanged.
#endif
#ifndef PBL_MIN_WAVES
#define PBL_MIN_WAVES 1
#endif
// SPLIT == 1: each of the WPB waves of a workgroup owns its own record (throughput mode, the
// L-layer stream).  SPLIT > 1 (== WPB): the waves of a workgroup COOPERATE on one record --
// wave w takes panels w, w+S, ... and salient rounds w, w+S, ... and the partial sums are
// merged through LDS -- so one small layer (256 records at N = 4096) still puts thousands of
// waves on the chip (latency mode, sequential decode).
// fp32 -> bf16 bits, round to nearest even; NaN stays NaN (csrc/pbl_act.hip)
__device__ __forceinline__ uint32_t bf16_bits_rne(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x0040u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// one bf16 (low 16 bits of b) -> the fp16 bits the kernel multiplies with: scaled by `down` (a power of two: exact) when the token is
// finite; else the indicator finite -> 0, +-inf -> +-1, NaN -> NaN (csrc/pbl_act.hip: act_bf16_prepare_kernel, the same arithmetic)
__device__ __forceinline__ uint32_t bf16_to_scaled_f16(uint32_t b, bool finite, float down) {
    const float f = __builtin_bit_cast(float, b << 16);
    float r;
    if (finite) r = f * down;
    else {
        const uint32_t a = (b << 16) & 0x7FFFFFFFu;
        r = a > 0x7F800000u ? f : (a == 0x7F800000u ? ((b & 0x8000u) ? -1.f : 1.f) : 0.f);
    }
    return uint32_t(__builtin_bit_cast(uint16_t, _Float16(r)));
}

// XB: bf16 activations converted in the staging phase, bf16 result (GemvArgs.x_bf16).  A template parameter, not a run-time branch:
// the fp16 instantiations -- the headline kernel among them -- are the code they were.
template <int MB, int WPB, bool SF, int SPLIT, bool XB = false>
__global__ __launch_bounds__(WPB * PBL_WAVE, (MB == 1 ? PBL_MIN_WAVES : 1)) void pbl_gemv_kernel(GemvArgs args) {
    static_assert(SPLIT == 1 || SPLIT == WPB, "split mode uses every wave of the workgroup on one record");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_amax[XB ? MB * WPB : 1];           // XB: per (token, wave) the largest |x| bit pattern
    __shared__ float s_tscale[XB ? MB : 1];                  //     per token the power of two (or +inf) the result is multiplied with
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    pbl_layer L;
    const _Float16* xg;
    void* yg;
    if (args.grouped && args.n_inl) {
        L = args.inl[blockIdx.y];
        xg = args.x_shared;
        yg = static_cast<char*>(args.y_shared) + args.inl_off[blockIdx.y] * (args.y_f32 ? 4 : 2);
    } else if (args.grouped) {
        L = args.layers[blockIdx.y];
        xg = args.x_shared ? args.x_shared : static_cast<const _Float16*>(args.xs[blockIdx.y]);
        yg = args.y_shared ? static_cast<char*>(args.y_shared) + args.y_off[blockIdx.y] * (args.y_f32 ? 4 : 2) : args.ys[blockIdx.y];
    } else {
        L = args.layer; xg = args.x; yg = args.y;
    }
    const size_t ldy = args.ldy ? size_t(args.ldy) : size_t(L.N);
    // XCD-aware record mapping (speed only, never correctness): the dispatcher places
    // workgroup b on XCD b % 8, so give each XCD a CONTIGUOUS range of a layer's records and
    // the 128-byte lines shared by neighbouring records stay within one L2.  Measured neutral
    // (FETCH_SIZE 1416.8 -> 1414.0 MB per 224-layer launch): there is no inter-record reuse.
    const uint32_t nwg = SPLIT > 1 ? L.NRB : (L.NRB + WPB - 1) / WPB;
    if (blockIdx.x >= nwg) return;  // whole workgroup exits together (grouped launches over-provision)
    const uint32_t xq = nwg >> 3, xr_ = nwg & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const uint32_t wg = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + xi;
    const uint32_t rb0 = SPLIT > 1 ? wg : wg * WPB;
    // fused push: the call number lives in this rank's buffer (last finished call + 1; the reduce kernel of the previous call has
    // published it: same stream); its parity picks the slot set
    int push_set = 0;
    if (args.push_world) {
        uint32_t q = __hip_atomic_load(pblp2p::ctl_ptr(args.push_peer[args.push_rank]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        q = q ? q : 1u;
        push_set = int(__builtin_amdgcn_readfirstlane(q) & 1u);
    }
    const int wslot = SPLIT > 1 ? wave : 0;      // this wave's first panel / salient round
    constexpr int WSTEP = SPLIT;                 // ... and its stride

    const int K = int(L.K), P = int(L.P);
    const int Kp = P * PBL_PANEL_COLS;
    const int xstride = Kp + 8;            // halves per token in LDS; [K, Kp+8) is zero
    _Float16* xs = reinterpret_cast<_Float16*>(smem);
    float2* part_all = reinterpret_cast<float2*>(smem + ((size_t(MB) * xstride * 2 + 15) & ~size_t(15)));

    // ---- record lookup + first loads, issued BEFORE x is staged so that HBM latency of
    //      the weight stream overlaps the L2->LDS copy of x and the barrier ----------------
    const uint32_t rb = SPLIT > 1 ? rb0 : rb0 + wave;
    const bool active = rb < L.NRB;
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[active ? rb : rb0];
    const uint8_t* rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    const int nfull = __builtin_amdgcn_readfirstlane(info.y);
    const int ntail = __builtin_amdgcn_readfirstlane(info.z);
    const int nexc = __builtin_amdgcn_readfirstlane(info.w);
    const int nch = nfull + ntail;
    const uint32_t tiles_off = PBL_TILES_OFF(L.G);
    const uint32_t off_sal = tiles_off + uint32_t(P) * 1024u;
    const uint32_t nchu = uint32_t(nch);

    const u32x4* tiles = reinterpret_cast<const u32x4*>(rec + tiles_off) + lane;
    const uint8_t* sal = rec + off_sal;
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const u32x4* codep = reinterpret_cast<const u32x4*>(sal + PBL_SAL_CODE_OFF(nchu));
    const uint8_t* tailcnt = sal + PBL_SAL_TAILCNT_OFF(nchu);

    // Three tile registers sets, statically indexed: a set is re-requested for the panel three steps on right after its last
    // use, so a request has two whole panels of arithmetic to land.  (Rounds 1-2 rotated t0 <- t1 <- t2: the copy touches the
    // register of a load issued at the top of the same iteration, hipcc waits vmcnt(0) in front of it, and the "two panels
    // ahead" prefetch was in fact one panel deep -- 1 KiB per wave in flight.)
    u32x4 t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
    (void)t2;                   // (only the three-set ring uses it)
    uint32_t s_c0 = 0;
    u32x4 s_d4 = {0, 0, 0, 0}, s_q4 = {0, 0, 0, 0};
    uint32_t abl = 0;
    (void)abl;
    if (active && PBL_ABLATE != 2) {
        if (wslot < P) t0 = load_tile(tiles + wslot * 64);
        if (wslot + WSTEP < P) t1 = load_tile(tiles + (wslot + WSTEP) * 64);
        if (PBL_TILE_RING == 3 && wslot + 2 * WSTEP < P) t2 = load_tile(tiles + (wslot + 2 * WSTEP) * 64);
        if (nch > wslot * PBL_WAVE) {
            const int c_first = wslot * PBL_WAVE + lane;
            const int cc = c_first < nch ? c_first : nch - 1;
            s_c0 = col0p[cc];
            s_d4 = load_sal(deltap + cc);
            s_q4 = load_sal(codep + cc);
        }
    }

    // ---- phase 0: stage x ---------------------------------------------------------
    if constexpr (XB) {
        // bf16 x: pass 1 finds every token's largest magnitude (bf16 << 16 is the fp32 pattern; |x| patterns order like unsigned
        // integers, inf / NaN on top), one barrier, pass 2 re-reads the row (L1 / L2) and writes the scaled fp16 copy
        const int nthr = WPB * PBL_WAVE;
        const uint16_t* xb = reinterpret_cast<const uint16_t*>(xg);
        const bool vec = (K & 7) == 0 && (reinterpret_cast<uintptr_t>(xb) & 15) == 0;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            uint32_t mx = 0;
            if (vec) {
                const u32x4* src = reinterpret_cast<const u32x4*>(xb + size_t(m) * K);
                for (int i = tid; i < (K >> 3); i += nthr) {
                    const u32x4 v = src[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t lo = (v[j] << 16) & 0x7FFFFFFFu, hi = v[j] & 0x7FFF0000u;
                        mx = lo > mx ? lo : mx;
                        mx = hi > mx ? hi : mx;
                    }
                }
            } else {
                for (int i = tid; i < K; i += nthr) {
                    const uint32_t a_ = (uint32_t(xb[size_t(m) * K + i]) << 16) & 0x7FFFFFFFu;
                    mx = a_ > mx ? a_ : mx;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = uint32_t(__shfl_xor(int(mx), off, PBL_WAVE));
                mx = o > mx ? o : mx;
            }
            if (lane == 0) s_amax[m * WPB + wave] = mx;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            uint32_t mx = 0;
#pragma unroll
            for (int w = 0; w < WPB; ++w) mx = s_amax[m * WPB + w] > mx ? s_amax[m * WPB + w] : mx;
            const bool finite = mx < 0x7F800000u;
            const int eb = int(mx >> 23) - 127 - 14;                     // 2^-e, e = max(0, exponent(amax) - 14): amax / 2^e < 2^15
            const int e = eb > 0 ? eb : 0;
            const float down = __builtin_bit_cast(float, uint32_t(127 - e) << 23);
            if (tid == 0) s_tscale[m] = finite ? __builtin_bit_cast(float, uint32_t(127 + e) << 23) : __builtin_inff();
            if (vec) {
                const u32x4* src = reinterpret_cast<const u32x4*>(xb + size_t(m) * K);
                u32x4* dst = reinterpret_cast<u32x4*>(xs + m * xstride);
                for (int i = tid; i < (K >> 3); i += nthr) {
                    const u32x4 v = src[i];
                    u32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = bf16_to_scaled_f16(v[j] & 0xFFFFu, finite, down) | (bf16_to_scaled_f16(v[j] >> 16, finite, down) << 16);
                    dst[i] = o;
                }
            } else {
                for (int i = tid; i < K; i += nthr)
                    reinterpret_cast<uint16_t*>(xs)[m * xstride + i] = uint16_t(bf16_to_scaled_f16(xb[size_t(m) * K + i], finite, down));
            }
            for (int i = K + tid; i < xstride; i += nthr) xs[m * xstride + i] = _Float16(0);
        }
    } else {
        const int nthr = WPB * PBL_WAVE;
        if ((K & 7) == 0 && (reinterpret_cast<uintptr_t>(xg) & 15) == 0) {
            const int nv = K >> 3;
            for (int m = 0; m < MB; ++m) {
                const u32x4* src = reinterpret_cast<const u32x4*>(xg + size_t(m) * K);
                u32x4* dst = reinterpret_cast<u32x4*>(xs + m * xstride);
                for (int i = tid; i < nv; i += nthr) dst[i] = src[i];
            }
        } else {
            for (int m = 0; m < MB; ++m)
                for (int i = tid; i < K; i += nthr) xs[m * xstride + i] = xg[size_t(m) * K + i];
        }
        for (int m = 0; m < MB; ++m)
            for (int i = K + tid; i < xstride; i += nthr) xs[m * xstride + i] = _Float16(0);
    }
    __syncthreads();
    if (!active) return;  // no barriers below this point

    // ---- phase 1: sign plane (tile loads run two panels ahead) ------------------------
    float acc[MB][16];
    float xl[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        xl[m] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    }
    // fp16x2 (1.0, 1.0) kept in a VGPR the compiler cannot constant-fold into a literal
    uint32_t c_one = 0x3C003C00u;
    asm volatile("" : "+v"(c_one));
    auto panel = [&](const u32x4& t, int p) {
        const uint32_t* xw = reinterpret_cast<const uint32_t*>(xs + p * PBL_PANEL_COLS) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t xr[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) xr[m] = xw[(m * xstride) / 2 + i * 64];
            if (PBL_ABLATE == 1) { abl ^= t[i] ^ xr[0]; continue; }
            word_step<MB>(t[i] ^ (PBL_ABLATE == 2 ? uint32_t(p * 4 + i + lane) : 0u), c_one, xr, acc, xl);
        }
    };
#if PBL_TILE_RING == 3
    for (int p = wslot; p < P; p += 3 * WSTEP) {
        panel(t0, p);
        if (p + 3 * WSTEP < P && PBL_ABLATE != 2) t0 = load_tile(tiles + (p + 3 * WSTEP) * 64);
        asm volatile("" ::: "memory");       // (keeps the next panel's LDS reads of x from being hoisted: 80 -> 64 VGPRs)
        if (p + WSTEP < P) {
            panel(t1, p + WSTEP);
            if (p + 4 * WSTEP < P && PBL_ABLATE != 2) t1 = load_tile(tiles + (p + 4 * WSTEP) * 64);
        }
        asm volatile("" ::: "memory");
        if (p + 2 * WSTEP < P) {
            panel(t2, p + 2 * WSTEP);
            if (p + 5 * WSTEP < P && PBL_ABLATE != 2) t2 = load_tile(tiles + (p + 5 * WSTEP) * 64);
        }
        asm volatile("" ::: "memory");
    }
#else
    for (int p = wslot; p < P; p += WSTEP) {              // the rotating form of rounds 1-2 (kept for A/B runs)
        u32x4 tn = t1;
        if (p + 2 * WSTEP < P && PBL_ABLATE != 2) tn = load_tile(tiles + (p + 2 * WSTEP) * 64);
        panel(t0, p);
        t0 = t1;
        t1 = tn;
    }
#endif

    // reduce the 16 row accumulators now: from here on one register per token carries the
    // binary result (lane l holds row rho(l), see transpose_reduce16)
    float accsum[MB], X[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        X[m] = wave_sum(xl[m]);
        accsum[m] = transpose_reduce16(acc[m], lane);
    }

    // row-owner lanes: lane l owns row rho(l) (4 lanes per row)
    const int rho = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    const int sub = lane & 3;
    const pbl_rowparams pr = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF)[rho];
    const pbl_rowinfo ri = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF)[rho];   // issued early: phase 3 must not wait on HBM
    // SF layers in a mixed grouped launch are told apart at run time (wave-uniform)
    const bool sf = SF && (L.flags & PBL_FLAG_SAL_F16);
    const uint8_t* crow = sal + PBL_SAL_CROW_OFF(nchu, uint32_t(ntail));
    const bool has_crow = L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16);

    // ---- phase 2: salient chunks (next round's loads issued before this round's math) ---
    float2* part = part_all + (SPLIT > 1 ? size_t(0) : size_t(wave) * ((PBL_PROBE & 1) ? (L.max_nch + 1) / 2 : L.max_nch) * MB);
    {
        const uint32_t xbase = uint32_t(reinterpret_cast<uintptr_t>(xs));      // LDS byte offset of x
        const uint32_t tok_stride = uint32_t(xstride) * 2u;
        const uint32_t zaddr = xbase + 2u * uint32_t(Kp);                      // a zero slot
        for (int base = wslot * PBL_WAVE; base < nch; base += WSTEP * PBL_WAVE) {
            const int c = base + lane;
            const bool valid = c < nch;
            const int cc = valid ? c : nch - 1;
            const uint32_t c0 = s_c0;
            const u32x4 d4 = s_d4, q4 = s_q4;
            if (base + WSTEP * PBL_WAVE < nch && PBL_ABLATE != 2) {
                const int cn = c + WSTEP * PBL_WAVE < nch ? c + WSTEP * PBL_WAVE : nch - 1;
                s_c0 = col0p[cn];
                s_d4 = load_sal(deltap + cn);
                s_q4 = load_sal(codep + cn);
            }
            float Q[MB], S[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) { Q[m] = 0.f; S[m] = 0.f; }
            int cnt = 16;
            if (base + PBL_WAVE > nfull) {
                if (cc >= nfull) cnt = tailcnt[cc - nfull];
                if (!valid) cnt = 0;
            }
            if (PBL_ABLATE == 1) {
                abl ^= c0 ^ d4[0] ^ d4[1] ^ d4[2] ^ d4[3] ^ q4[0] ^ q4[1] ^ q4[2] ^ q4[3];
            } else if (SF && sf) {
                // this chunk's row -> its (scale, zero) from the row-owner lane (no memory round trip)
                const int crw = crow[cc];
                const int owner = ((crw & 8) << 2) | ((crw & 4) << 2) | ((crw & 2) << 2) | ((crw & 1) << 2);
                const float ss = __shfl(pr.sscale, owner, PBL_WAVE), sz = __shfl(pr.szero, owner, PBL_WAVE);
                if (base + PBL_WAVE <= nfull)
                    chunk_accumulate<MB, false, true>(xbase, tok_stride, c0, d4, q4, 16, zaddr, c_one, ss, sz, Q, S);
                else
                    chunk_accumulate<MB, true, true>(xbase, tok_stride, c0, d4, q4, cnt, zaddr, c_one, ss, sz, Q, S);
            } else if (base + PBL_WAVE <= nfull) {
                chunk_accumulate<MB, false, false>(xbase, tok_stride, c0, d4, q4, 16, zaddr, c_one, 0.f, 0.f, Q, S);
            } else {
                chunk_accumulate<MB, true, false>(xbase, tok_stride, c0, d4, q4, cnt, zaddr, c_one, 0.f, 0.f, Q, S);
            }
            if (valid) {
#pragma unroll
                for (int m = 0; m < MB; ++m)   // code mode: undo the 1024 code bias here, once per chunk
                    part[size_t(m) * L.max_nch + ((PBL_PROBE & 1) ? (c >> 1) : c)] = make_float2((SF && sf) ? Q[m] : fmaf(-1024.f, S[m], Q[m]), S[m]);
            }
        }
    }
    if constexpr (SPLIT > 1) {
        // merge the S waves' binary partials (lane l of every wave holds row rho(l)); the chunk
        // partials already sit in the shared `part` array
        float2* red = part_all + size_t(L.max_nch) * MB;
#pragma unroll
        for (int m = 0; m < MB; ++m) red[(m * SPLIT + wave) * PBL_WAVE + lane] = make_float2(accsum[m], X[m]);
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float a_ = 0.f, x_ = 0.f;
#pragma unroll
            for (int w = 0; w < SPLIT; ++w) {
                const float2 v = red[(m * SPLIT + w) * PBL_WAVE + lane];
                a_ += v.x; x_ += v.y;
            }
            accsum[m] = a_; X[m] = x_;
        }
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- phase 3: reduce, combine, store -------------------------------------------
    float A, B;
    class_consts(rho & 7, A, B);
    const float alpha = 0.5f * (pr.hi - pr.lo), mu = 0.5f * (pr.hi + pr.lo);
    const uint32_t row = rb * 16 + rho;
    // exceptions are read as ONE aligned 64-bit word each: a scalar load of the fp32
    // field off a 2-byte-aligned base silently drops the low address bits on gfx950
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));

#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float Q = 0.f, S = 0.f;
        const float2* pm = part + size_t(m) * L.max_nch;
        for (int k = sub; k < int(ri.nfull); k += 4) {
            const float2 v = pm[(PBL_PROBE & 1) ? ((ri.start + k) >> 1) : (ri.start + k)];
            Q += v.x; S += v.y;
        }
        for (int k = sub; k < int(ri.ntail); k += 4) {
            const float2 v = pm[(PBL_PROBE & 1) ? ((nfull + ri.tailidx + k) >> 1) : (nfull + ri.tailidx + k)];
            Q += v.x; S += v.y;
        }
        Q = quad_sum(Q); S = quad_sum(S);
        float e = 0.f;
        for (int k = 0; k < nexc; ++k) {
            const uint2 ex = exc[k];  // {col | row << 16, value}
            if (int(ex.x >> 16) == rho)
                e += (__builtin_bit_cast(float, ex.y) - pr.hi) * float(xs[m * xstride + (ex.x & 0xFFFFu)]);
        }
        // explicit fma chain: every template instantiation rounds identically (M=4 == 4 x M=1)
        const float D = fmaf(A, accsum[m], -(B * X[m]));
        const float sal = (SF && sf) ? fmaf(-pr.hi, S, Q) : fmaf(pr.sscale, fmaf(-pr.szero, S, Q), -(pr.hi * S));
        float yv = fmaf(alpha, D, fmaf(mu, X[m], sal)) + e;
        if constexpr (XB) {                                      // back to the token's own scale, then the bias: pbl_act_finish's fma
            const float bv = (L.bias && row < L.N) ? L.bias[row] : 0.f;
            yv = fmaf(yv, s_tscale[m], bv);
        } else if (L.bias && row < L.N) yv += L.bias[row];
        if (PBL_ABLATE == 1 && abl == 0x9E3779B9u) yv += 1.f;
        if (sub == 0 && row < L.N) {
            if (args.push_world) {
                for (int p = 0; p < args.push_world; ++p)       // 7 xGMI stores + 1 local, 4 bytes each from 16 lanes (64-byte runs)
                    pblp2p::slot_ptr(args.push_peer[p], push_set, args.push_rank, args.push_world, args.push_cap)[size_t(m) * ldy + row] = yv;
            } else if (args.y_f32) static_cast<float*>(yg)[size_t(m) * ldy + row] = yv;
            else if constexpr (XB) static_cast<uint16_t*>(yg)[size_t(m) * ldy + row] = uint16_t(bf16_bits_rne(yv));
            else static_cast<_Float16*>(yg)[size_t(m) * ldy + row] = _Float16(yv);
        }
    }
    if (args.push_world) {
        // this record's rows are visible system wide, THEN it counts itself at every rank (program order of one wave: the
        // fence drains the stores above before the atomics below are issued)
        __threadfence_system();
        if (lane == 0)
            for (int p = 0; p < args.push_world; ++p)
                __hip_atomic_fetch_add(pblp2p::count_ptr(args.push_peer[p], push_set, args.push_rank), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#ifdef PBL_PROTO_ROWMAJOR   // experimental build only: the row-major-plane GEMV prototype lives in tools/ (see the file's header)
#define PBL_PROTO_SECTION_KERNEL
#include "../../tools/proto_rowmajor_kernel.inc"
#undef PBL_PROTO_SECTION_KERNEL
#endif

size_t lds_bytes(uint32_t P, uint32_t max_nch, int mb, int wpb, int split = 1) {
    const size_t xstride = size_t(P) * PBL_PANEL_COLS + 8;
    size_t s = (size_t(mb) * xstride * 2 + 15) & ~size_t(15);
    if (split > 1) s += (size_t(max_nch) + size_t(split) * PBL_WAVE) * mb * sizeof(float2);
    else s += size_t(wpb) * ((PBL_PROBE & 1) ? (max_nch + 1) / 2 : max_nch) * mb * sizeof(float2);
    return s + 16;
}

template <int MB, int WPB, bool SF, int SPLIT = 1>
int launch(const GemvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    auto k = a.x_bf16 ? pbl_gemv_kernel<MB, WPB, SF, SPLIT, true> : pbl_gemv_kernel<MB, WPB, SF, SPLIT, false>;
    // the bf16 instantiation holds s_amax / s_tscale in STATIC LDS on top of the dynamic size (4 MB (WPB + 1) bytes, rounded up):
    // both limits are on the sum (a layer just under a limit failed the launch instead of answering UNSUPPORTED, which
    // pbl_linear_bf16's callers turn into the three-launch form)
    const size_t stat = a.x_bf16 ? ((size_t(4) * MB * (WPB + 1) + 15) & ~size_t(15)) + 16 : 16;
    if (lds + stat > 64 * 1024) {
        if (lds + stat > 160 * 1024) return PBL_ERR_UNSUPPORTED;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)) != hipSuccess)
            return a.x_bf16 ? PBL_ERR_UNSUPPORTED : PBL_ERR_LAUNCH;
    }
    GemvArgs args = a;
    void* argv[] = {&args};
    return hipLaunchKernel(reinterpret_cast<const void*>(k), grid, dim3(WPB * PBL_WAVE), argv, lds, st) == hipSuccess
               ? PBL_OK : PBL_ERR_LAUNCH;
}


// ---------------------------------------------------------------------------------------
// Column-group variant (G > 1: per-(row, group) hi/lo, gptq_pb --groupsize 128).  Same
// record format and bit classes; the 16 accumulators are flushed into per-row totals at
// every group boundary with per-(row,group) coefficients  ca = alpha*A_c,  cb = mu - alpha*B_c
// kept in LDS, and the salient correction looks hi up per entry.  A correctness-first
// kernel for the non-default configuration; the G == 1 kernel above is the tuned one.
template <int MB, int WPB>
__global__ __launch_bounds__(WPB * PBL_WAVE) void pbl_gemv_groups_kernel(GemvArgs args, int gshift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    pbl_layer L;
    const _Float16* xg;
    void* yg;
    if (args.grouped && args.n_inl) {
        L = args.inl[blockIdx.y];
        xg = args.x_shared;
        yg = static_cast<char*>(args.y_shared) + args.inl_off[blockIdx.y] * (args.y_f32 ? 4 : 2);
    } else if (args.grouped) {
        L = args.layers[blockIdx.y];
        xg = args.x_shared ? args.x_shared : static_cast<const _Float16*>(args.xs[blockIdx.y]);
        yg = args.y_shared ? static_cast<char*>(args.y_shared) + args.y_off[blockIdx.y] * (args.y_f32 ? 4 : 2) : args.ys[blockIdx.y];
    } else {
        L = args.layer; xg = args.x; yg = args.y;
    }
    const size_t ldy = args.ldy ? size_t(args.ldy) : size_t(L.N);
    xg += size_t(args.tok0) * L.K;
    yg = static_cast<char*>(yg) + size_t(args.tok0) * ldy * (args.y_f32 ? 4 : 2);
    const uint32_t rb0 = blockIdx.x * WPB;
    if (rb0 >= L.NRB) return;
    const int K = int(L.K), P = int(L.P), G = int(L.G);
    const int Kp = P * PBL_PANEL_COLS, xstride = Kp + 8;
    const int gw = (K / G) / 128;  // dwords of a lane per column group
    if (gshift < 0) gshift = G > 1 ? 31 - __builtin_clz(uint32_t(K / G)) : 31;   // grouped launch: every layer has its own group size (G == 1: one group)
    _Float16* xs = reinterpret_cast<_Float16*>(smem);
    char* after_x = smem + ((size_t(MB) * xstride * 2 + 15) & ~size_t(15));
    float2* cacb = reinterpret_cast<float2*>(after_x) + size_t(wave) * G * 16;          // [g][rho]
    float* hitab = reinterpret_cast<float*>(after_x + size_t(WPB) * G * 16 * 8) + size_t(wave) * G * 16;
    char* after_tab = after_x + size_t(WPB) * G * 16 * 12;
    float2* part = reinterpret_cast<float2*>(after_tab) + size_t(wave) * L.max_nch * MB;
    float* partH = reinterpret_cast<float*>(after_tab + size_t(WPB) * L.max_nch * MB * 8) + size_t(wave) * L.max_nch * MB;

    const uint32_t rb = rb0 + wave;
    const bool active = rb < L.NRB;
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[active ? rb : rb0];
    const uint8_t* rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    const int nfull = __builtin_amdgcn_readfirstlane(info.y);
    const int ntail = __builtin_amdgcn_readfirstlane(info.z);
    const int nexc = __builtin_amdgcn_readfirstlane(info.w);
    const int nch = nfull + ntail;
    const uint32_t tiles_off = PBL_TILES_OFF(uint32_t(G));
    const uint32_t off_sal = tiles_off + uint32_t(P) * 1024u;

    {   // stage x, build this wave's coefficient tables
        const int nthr = WPB * PBL_WAVE;
        for (int m = 0; m < MB; ++m) {
            for (int i = tid; i < K; i += nthr) xs[m * xstride + i] = xg[size_t(m) * K + i];
            for (int i = K + tid; i < xstride; i += nthr) xs[m * xstride + i] = _Float16(0);
        }
        if (active) {
            const float2* ghl = reinterpret_cast<const float2*>(rec + PBL_REC_GHL_OFF);
            const pbl_rowparams* rp = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
            for (int idx = lane; idx < 16 * G; idx += PBL_WAVE) {
                const int rho = idx & 15, g = idx >> 4;
                // a group-free layer riding along in a grouped launch has no ghl table: its one "group" is the row's own levels
                const float2 hl = G > 1 ? ghl[rho * G + g] : make_float2(rp[rho].hi, rp[rho].lo);
                float A, B;
                class_consts(rho & 7, A, B);
                const float al = 0.5f * (hl.x - hl.y), mu = 0.5f * (hl.x + hl.y);
                cacb[idx] = make_float2(al * A, fmaf(-al, B, mu));
                hitab[idx] = hl.x;
            }
        }
    }
    __syncthreads();
    if (!active) return;

    float acc[MB][16], tot[MB][16], xl[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        xl[m] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[m][r] = 0.f; tot[m][r] = 0.f; }
    }
    uint32_t c_one = 0x3C003C00u;
    asm volatile("" : "+v"(c_one));
    const u32x4* tiles = reinterpret_cast<const u32x4*>(rec + tiles_off) + lane;
    int wi = 0, g = 0;
    for (int p = 0; p < P; ++p) {
        const u32x4 t = __builtin_nontemporal_load(tiles + p * 64);
        const uint32_t* xw = reinterpret_cast<const uint32_t*>(xs + p * PBL_PANEL_COLS) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t xr[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) xr[m] = xw[(m * xstride) / 2 + i * 64];
            word_step<MB>(t[i], c_one, xr, acc, xl);
            if (++wi == gw) {
                if (g < G) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float2 c = cacb[g * 16 + r];
#pragma unroll
                        for (int m = 0; m < MB; ++m)
                            tot[m][r] = fmaf(c.y, xl[m], fmaf(c.x, acc[m][r], tot[m][r]));
                    }
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    xl[m] = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
                }
                wi = 0;
                ++g;
            }
        }
    }
    float T[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) T[m] = transpose_reduce16(tot[m], lane);

    // salient chunks: per entry  q*x, x and hi[row][group(col)]*x
    const uint8_t* sal = rec + off_sal;
    const uint32_t nchu = uint32_t(nch);
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const u32x4* codep = reinterpret_cast<const u32x4*>(sal + PBL_SAL_CODE_OFF(nchu));
    const uint8_t* tailcnt = sal + PBL_SAL_TAILCNT_OFF(nchu);
    const uint8_t* crow = sal + PBL_SAL_CROW_OFF(nchu, uint32_t(ntail));
    const bool has_crow = (L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16)) != 0;
    const pbl_rowinfo* rinfo = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF);
    for (int base = 0; base < nch; base += PBL_WAVE) {
        const int c = base + lane;
        if (c < nch) {
            const u32x4 d4 = deltap[c], q4 = codep[c];
            const int cnt = c >= nfull ? int(tailcnt[c - nfull]) : 16;
            int row = 0;
            if (has_crow) row = crow[c];
            else {            // (group-free member without per-chunk row ids: the row whose chunk range holds c)
                for (int r = 0; r < 16; ++r) {
                    const pbl_rowinfo q = rinfo[r];
                    const bool in = c < nfull ? (c >= int(q.start) && c < int(q.start) + int(q.nfull))
                                              : (c - nfull >= int(q.tailidx) && c - nfull < int(q.tailidx) + int(q.ntail));
                    row = in ? r : row;
                }
            }
            const pbl_rowparams cp = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF)[row];
            const bool sf16 = L.flags & PBL_FLAG_SAL_F16;
            uint32_t col2 = 2u * col0p[c];
            float Q[MB], S[MB], H[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) { Q[m] = 0.f; S[m] = 0.f; H[m] = 0.f; }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                col2 += (d4[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                if (k < cnt) {
                    const uint32_t col = col2 >> 1;
                    float qf = cp.sscale * (float((q4[k >> 2] >> (8 * (k & 3))) & 0xFFu) - cp.szero);  // the weight itself
                    if (sf16) qf = float(round_f16_twice(qf));
                    const float hv = hitab[(col >> gshift) * 16 + row];
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const float xv = float(xs[m * xstride + col]);
                        Q[m] = fmaf(qf, xv, Q[m]);
                        S[m] += xv;
                        H[m] = fmaf(hv, xv, H[m]);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                part[size_t(m) * L.max_nch + c] = make_float2(Q[m], S[m]);
                partH[size_t(m) * L.max_nch + c] = H[m];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const int rho = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    const int sub = lane & 3;
    const pbl_rowinfo ri = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF)[rho];
    const uint32_t row = rb * 16 + rho;
    const uint2* exc = reinterpret_cast<const uint2*>(sal + (has_crow ? PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), true)
                                                                       : PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), false)));
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float Q = 0.f, S = 0.f, H = 0.f;
        const float2* pm = part + size_t(m) * L.max_nch;
        const float* ph = partH + size_t(m) * L.max_nch;
        for (int k = sub; k < int(ri.nfull); k += 4) {
            const float2 v = pm[ri.start + k];
            Q += v.x; S += v.y; H += ph[ri.start + k];
        }
        for (int k = sub; k < int(ri.ntail); k += 4) {
            const float2 v = pm[nfull + ri.tailidx + k];
            Q += v.x; S += v.y; H += ph[nfull + ri.tailidx + k];
        }
        Q = quad_sum(Q); S = quad_sum(S); H = quad_sum(H);
        float e = 0.f;
        for (int k = 0; k < nexc; ++k) {
            const uint2 ex = exc[k];
            if (int(ex.x >> 16) == rho) {
                const uint32_t col = ex.x & 0xFFFFu;
                e += (__builtin_bit_cast(float, ex.y) - hitab[(col >> gshift) * 16 + rho]) * float(xs[m * xstride + col]);
            }
        }
        float yv = T[m] + (Q - H) + e;
        if (L.bias && row < L.N) yv += L.bias[row];
        if (sub == 0 && row < L.N) {
            if (args.y_f32) static_cast<float*>(yg)[size_t(m) * ldy + row] = yv;
            else static_cast<_Float16*>(yg)[size_t(m) * ldy + row] = _Float16(yv);
        }
    }
}


// ---------------------------------------------------------------------------------------
// Device-side unpack: PBL1 -> dense row-major [N,K] (fp16 or fp32); the GEMM-regime path
// (the dense matrix lives only in a transient workspace).
template <typename OutT>
__global__ __launch_bounds__(4 * PBL_WAVE) void pbl_unpack_kernel(pbl_layer L, OutT* __restrict__ Wout, int seg_panels) {
    // A workgroup owns 4 rows of a record (grid = 4 * NRB), one per wave; a wave builds its output row in LDS, in
    // column segments of seg_panels * 512 columns -- sign plane expanded, then the row's salient chunks
    // (column-sorted) and exceptions written over it -- and streams the finished segment out with contiguous
    // 16-byte stores.  No global scatter: every HBM line is written whole, once.
    extern __shared__ __attribute__((aligned(16))) char smem_u[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t rb = blockIdx.x >> 2;
    const int row0 = int(blockIdx.x & 3) * 4;
    const int K = int(L.K), P = int(L.P), G = int(L.G);
    const uint8_t* blob = static_cast<const uint8_t*>(L.blob);
    const uint4 info = reinterpret_cast<const uint4*>(blob + sizeof(pbl_blob_header))[rb];
    const uint8_t* rec = blob + size_t(__builtin_amdgcn_readfirstlane(info.x)) * 16;
    const int nfull = __builtin_amdgcn_readfirstlane(info.y), ntail = __builtin_amdgcn_readfirstlane(info.z);
    const int nexc = __builtin_amdgcn_readfirstlane(info.w), nch = nfull + ntail;
    const bool groups = L.flags & PBL_FLAG_HAS_GROUPS, sf16 = L.flags & PBL_FLAG_SAL_F16;
    const uint32_t tiles_off = PBL_TILES_OFF(uint32_t(G));
    const pbl_rowparams* params = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
    const pbl_rowinfo* rinfo = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF);
    const float2* ghl = reinterpret_cast<const float2*>(rec + PBL_REC_GHL_OFF);
    const int gwords = groups ? (K / G) / 128 : (1 << 30);   // dwords of a lane per column group
    const int nrows = (L.N - rb * 16) < 16u ? int(L.N - rb * 16) : 16;
    const u32x4* tiles = reinterpret_cast<const u32x4*>(rec + tiles_off) + lane;
    const uint8_t* sal = rec + tiles_off + uint32_t(P) * 1024u;
    const uint32_t nchu = uint32_t(nch);
    const uint16_t* col0p = reinterpret_cast<const uint16_t*>(sal);
    const u32x4* deltap = reinterpret_cast<const u32x4*>(sal + PBL_SAL_DELTA_OFF(nchu));
    const u32x4* codep = reinterpret_cast<const u32x4*>(sal + PBL_SAL_CODE_OFF(nchu));
    const uint8_t* tailcnt = sal + PBL_SAL_TAILCNT_OFF(nchu);
    const bool has_crow = L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16);
    const uint2* exc = reinterpret_cast<const uint2*>(sal + PBL_SAL_EXC_OFF(nchu, uint32_t(ntail), has_crow));
    const int seg_cols = seg_panels * PBL_PANEL_COLS;
    OutT* rowbuf = reinterpret_cast<OutT*>(smem_u) + size_t(wave) * seg_cols;   // this wave's row segment under construction
    const bool vec_ok = (size_t(K) * sizeof(OutT)) % 16 == 0;                    // every row starts 16-byte aligned

    const int rho = row0 + wave;
    if (rho < nrows) {
        const int pos = rho < 8 ? rho + 8 : rho - 8;
        const pbl_rowparams pr = params[rho];
        const pbl_rowinfo ri = rinfo[rho];
        const int n_full = int(ri.nfull), n_all = n_full + int(ri.ntail);
        for (int p0 = 0; p0 < P; p0 += seg_panels) {
            const int p1 = p0 + seg_panels < P ? p0 + seg_panels : P;
            const uint32_t c_lo = uint32_t(p0) * PBL_PANEL_COLS, c_n = uint32_t(p1 - p0) * PBL_PANEL_COLS;
            // 1. sign plane of the row: lane l owns columns 512p + 128i + 2l + {0,1}
            for (int p = p0; p < p1; ++p) {
                const u32x4 t = tiles[p * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float hi = pr.hi, lo = pr.lo;
                    if (groups) { const int g = (p * 4 + i) / gwords; const float2 hl = ghl[rho * G + (g < G ? g : G - 1)]; hi = hl.x; lo = hl.y; }
                    OutT* d = rowbuf + (p - p0) * PBL_PANEL_COLS + i * 128 + 2 * lane;
                    d[0] = OutT(((t[i] >> pos) & 1u) ? hi : lo);
                    d[1] = OutT(((t[i] >> (16 + pos)) & 1u) ? hi : lo);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 2. the row's salient chunks (full chunks [start, start + nfull), then its tail chunks); entries outside
            //    the segment are skipped
            for (int j = lane; j < n_all; j += PBL_WAVE) {
                const int c = j < n_full ? int(ri.start) + j : nfull + int(ri.tailidx) + (j - n_full);
                const int cnt = c >= nfull ? int(tailcnt[c - nfull]) : 16;
                const u32x4 d4 = deltap[c], q4 = codep[c];
                uint32_t col = uint32_t(col0p[c]) - c_lo;                 // wraps for columns left of the segment
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    col += ((d4[e >> 2] >> (8 * (e & 3))) & 0xFFu) >> 1;
                    float w = pr.sscale * (float((q4[e >> 2] >> (8 * (e & 3))) & 0xFFu) - pr.szero);
                    if (sf16) w = float(round_f16_twice(w));
                    if (e < cnt && col < c_n) rowbuf[col] = OutT(w);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 3. exceptions of the row (explicit values, written last)
            for (int k = lane; k < nexc; k += PBL_WAVE) {
                const uint2 ex = exc[k];
                const uint32_t col = (ex.x & 0xFFFFu) - c_lo;
                if (int(ex.x >> 16) == rho && col < c_n) rowbuf[col] = OutT(__builtin_bit_cast(float, ex.y));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 4. stream the segment out
            OutT* dst = Wout + size_t(rb * 16 + rho) * K + c_lo;
            const int ncol = int(c_lo + c_n) <= K ? int(c_n) : K - int(c_lo);
            if (vec_ok) {
                constexpr int V = 16 / sizeof(OutT);
                for (int j = lane * V; j < ncol; j += PBL_WAVE * V) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(rowbuf + j);
                    if (PBL_UNPACK_NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + j));
                    else *reinterpret_cast<u32x4*>(dst + j) = v;
                }
            } else {
                for (int j = lane; j < ncol; j += PBL_WAVE) dst[j] = rowbuf[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

size_t lds_bytes_groups(uint32_t P, uint32_t G, uint32_t max_nch, int mb, int wpb) {
    const size_t xstride = size_t(P) * PBL_PANEL_COLS + 8;
    size_t s = (size_t(mb) * xstride * 2 + 15) & ~size_t(15);
    s += size_t(wpb) * G * 16 * 12;
    s += size_t(wpb) * max_nch * mb * 12;
    return s + 16;
}

template <int MB>
int launch_groups(const GemvArgs& a, int gshift, dim3 grid, size_t lds, hipStream_t st) {
    auto k = pbl_gemv_groups_kernel<MB, 1>;
    if (lds > 160 * 1024) return PBL_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess)
        return PBL_ERR_LAUNCH;
    GemvArgs args = a;
    void* argv[] = {&args, &gshift};
    return hipLaunchKernel(reinterpret_cast<const void*>(k), grid, dim3(PBL_WAVE), argv, lds, st) == hipSuccess
               ? PBL_OK : PBL_ERR_LAUNCH;
}

int linear_groups(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, hipStream_t st) {
    const uint32_t gs = layer->K / layer->G;
    if (layer->K % layer->G || gs % 128 || (gs & (gs - 1))) return PBL_ERR_UNSUPPORTED;  // power-of-two groups
    int gshift = 0;
    while ((1u << gshift) < gs) ++gshift;
    const size_t esz = y_f32 ? 4 : 2;
    int mb_max = 2;   // two tokens per weight pass keeps the 2x16 accumulators in registers
    while (mb_max > 1 && lds_bytes_groups(layer->P, layer->G, layer->max_nch, mb_max, 1) > 96 * 1024) --mb_max;
    for (int m0 = 0; m0 < M; m0 += mb_max) {
        const int mb = M - m0 < mb_max ? M - m0 : mb_max;
        GemvArgs a{};
        a.layer = *layer;
        a.x = static_cast<const _Float16*>(x) + size_t(m0) * layer->K;
        a.y = static_cast<char*>(y) + size_t(m0) * layer->N * esz;
        a.M = mb; a.y_f32 = y_f32; a.grouped = 0;
        const dim3 grid(layer->NRB, 1, 1);
        const size_t lds = lds_bytes_groups(layer->P, layer->G, layer->max_nch, mb, 1);
        const int rc = mb == 1 ? launch_groups<1>(a, gshift, grid, lds, st) : launch_groups<2>(a, gshift, grid, lds, st);
        if (rc != PBL_OK) return rc;
    }
    return PBL_OK;
}

// Grouped / fused launch of layers that carry column groups (groupsize 128 / 256 / ...: gptq_pb/run_all.sh): the column-group
// kernel, one wave per record, grid (records, layers); every layer's group size must be a power of two (each layer decodes
// its own from K / G) and every K a multiple of 128.  Group-free layers may ride along (they are one group).
int grouped_groups(const GemvArgs& a, int M, uint32_t max_NRB, uint32_t max_K, uint32_t max_nch, hipStream_t st) {
    if (max_K & 127) return PBL_ERR_UNSUPPORTED;
    const uint32_t P = (max_K + PBL_PANEL_COLS - 1) / PBL_PANEL_COLS, Gmax = max_K / 128;
    const dim3 grid(max_NRB, a.Lc, 1);
    for (int m0 = 0; m0 < M; m0 += 2) {                       // two tokens per weight pass (the 2 x 16 accumulators stay in registers)
        const int mb = M - m0 < 2 ? M - m0 : 2;
        GemvArgs b = a;
        b.M = mb; b.tok0 = m0;
        const size_t lds = lds_bytes_groups(P, Gmax, max_nch, mb, 1);
        const int rc = mb == 1 ? launch_groups<1>(b, -1, grid, lds, st) : launch_groups<2>(b, -1, grid, lds, st);
        if (rc != PBL_OK) return rc;
    }
    return PBL_OK;
}

template <int WPB, bool SF>
int launch_mb2(int mb, const GemvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    switch (mb) {
        case 1: return launch<1, WPB, SF>(a, grid, lds, st);
        case 2: return launch<2, WPB, SF>(a, grid, lds, st);
        case 3: return launch<3, WPB, SF>(a, grid, lds, st);
        case 4: return launch<4, WPB, SF>(a, grid, lds, st);
        default: return PBL_ERR_INVALID_ARG;
    }
}
template <int WPB>
int launch_mb(int mb, bool sf, const GemvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    return sf ? launch_mb2<WPB, true>(mb, a, grid, lds, st) : launch_mb2<WPB, false>(mb, a, grid, lds, st);
}
// latency mode: S waves per record
template <int S>
int launch_split(int mb, bool sf, const GemvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    if (sf) {
        switch (mb) {
            case 1: return launch<1, S, true, S>(a, grid, lds, st);
            case 2: return launch<2, S, true, S>(a, grid, lds, st);
            case 3: return launch<3, S, true, S>(a, grid, lds, st);
            case 4: return launch<4, S, true, S>(a, grid, lds, st);
        }
    } else {
        switch (mb) {
            case 1: return launch<1, S, false, S>(a, grid, lds, st);
            case 2: return launch<2, S, false, S>(a, grid, lds, st);
            case 3: return launch<3, S, false, S>(a, grid, lds, st);
            case 4: return launch<4, S, false, S>(a, grid, lds, st);
        }
    }
    return PBL_ERR_INVALID_ARG;
}
#ifndef PBL_SPLIT_TARGET_WAVES
#define PBL_SPLIT_TARGET_WAVES 8192u
#endif
// waves per record so that a launch of `records` records fields about a full chip of waves
// (256 CUs x 32); more waves per record shorten every wave's serial load -> compute chain
int pick_split(uint32_t records, uint32_t P) {
    int s = 1;
    while (s < 8 && records * uint32_t(s) < PBL_SPLIT_TARGET_WAVES && uint32_t(2 * s) <= (P > 1 ? P : 1u) * 2u) s *= 2;
    return s;
}

}  // namespace

extern "C" {

size_t pbl_gemv_lds_bytes(const pbl_layer* layer, int m) {
    if (!layer || m < 1 || m > PBL_MAX_TOKENS_PER_LAUNCH) return 0;
    return lds_bytes(layer->P, layer->max_nch, m, 4);
}

int pbl_linear_f16(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* stream) {
    return pbl_linear_f16_ws(layer, x, y, M, y_f32, nullptr, 0, stream);
}

// Routing of one layer's forward between the GEMV (weights streamed once per `mb` tokens) and the matrix-core kernel (once per
// 32 tokens, but with a per-record expansion cost that only pays off from a handful of tokens on).  Measured on MI355X
// (tools/bench_route.py): the GEMV wins while all tokens fit ONE pass whose LDS footprint (x tile + chunk partials) leaves at
// least two workgroups per CU; a pass squeezed into a single workgroup per CU (K = 11008 with 3 tokens, 13824x5120 at 20 %
// salients with 4) is 3-10x slower than the matrix-core kernel, and so is the matrix-core kernel without its K split.
struct Route {
    int split, wpb, mb_max;   // GEMV: waves per record, waves per workgroup, tokens per pass
    bool mfma;                // more tokens than one GEMV pass takes and the matrix-core kernel applies
};
#ifndef PBL_GEMV_LDS_BUDGET
#define PBL_GEMV_LDS_BUDGET (80 * 1024)   // two workgroups per CU
#endif
static Route route_of(const pbl_layer* layer, int M, bool x_aligned) {
    Route r;
    if (layer->G != 1) {
        r.split = 1; r.wpb = 1; r.mb_max = 2;     // the column-group GEMV keeps 2 x 16 accumulators in registers
        while (r.mb_max > 1 && lds_bytes_groups(layer->P, layer->G, layer->max_nch, r.mb_max, 1) > 96 * 1024) --r.mb_max;
    } else {
        // throughput mode (a wave per record, 4 per workgroup) when there are enough records to fill the
        // chip; otherwise latency mode: S waves share a record
        r.split = layer->NRB >= PBL_SPLIT_TARGET_WAVES ? 1 : pick_split(layer->NRB, layer->P);
        r.wpb = r.split > 1 ? r.split : (layer->NRB >= 1024 ? 4 : 1);
        r.mb_max = PBL_MAX_TOKENS_PER_LAUNCH;
        while (r.mb_max > 1 && lds_bytes(layer->P, layer->max_nch, r.mb_max, r.wpb, r.split) > PBL_GEMV_LDS_BUDGET) --r.mb_max;
    }
    // (tiny layers at M <= 8 stay on the GEMV: both are launch-latency bound and the GEMV starts faster)
    r.mfma = M > r.mb_max && (M > 8 || layer->NRB >= 128) && !(layer->K & 7) && x_aligned &&
             (layer->flags & PBL_FLAG_TAIL_REPEAT) && (layer->flags & PBL_FLAG_SLABS);
    if (r.mfma && M <= 8 && layer->G == 1) {
        // a few tokens more than one pass holds: several GEMV passes can still beat the matrix-core kernel.  Estimates in us,
        // fitted to tools/bench_route.py on MI355X (4096^2, 11008x4096, 4096x11008, 13824x5120 at 5-20 % salients; within 20 %):
        //   GEMV pass of mb tokens   3.5 + packed bytes / 6.5 TB/s + (N K / 1e6) * (0.04 + 0.33 * salient density) * mb
        //   matrix-core kernel       8.5 + (N K / 1e6) * (0.27 + 0.7 * salient density)      (flat in M up to 16 tokens)
        const double wm = double(layer->N) * layer->K * 1e-6;
        const double nnz = double(layer->max_nch) * layer->NRB * 16.0;             // upper bound: every record as full as the fullest
        const double bytes = double(layer->N) * layer->K / 8 + nnz * 2.125;
        const double pass0 = 3.5 + bytes / 6.5e6, per_tok = wm * (0.04 + 0.33 * nnz / (wm * 1e6));
        double gemv = 0;
        for (int m0 = 0; m0 < M; m0 += r.mb_max) gemv += pass0 + per_tok * (M - m0 < r.mb_max ? M - m0 : r.mb_max);
        r.mfma = 8.5 + wm * (0.27 + 0.7 * nnz / (wm * 1e6)) < gemv;
    }
    if (r.mfma && layer->G != 1) {
        const uint32_t gs = layer->K / layer->G;
        r.mfma = gs * layer->G == layer->K && !(gs & (gs - 1)) && gs >= 128;
    }
    return r;
}

size_t pbl_linear_workspace_bytes(const pbl_layer* layer, int M) {
    if (!layer || M < 1) return 0;
    return route_of(layer, M, true).mfma ? pbl_mfma_workspace_bytes(layer, M < 32 ? M : 32) : 0;
}

static int mfma_passes(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* workspace, size_t workspace_bytes, void* stream) {
    const size_t esz = y_f32 ? 4 : 2;
    int rc = PBL_OK;
    for (int m0 = 0; m0 < M && rc == PBL_OK; m0 += 32) {
        const int mb = M - m0 < 32 ? M - m0 : 32;
        rc = pbl_gemm_mfma_f16_ws(layer, static_cast<const _Float16*>(x) + size_t(m0) * layer->K,
                                  static_cast<char*>(y) + size_t(m0) * layer->N * esz, mb, y_f32, workspace, workspace_bytes, stream);
    }
    return rc;
}

int pbl_linear_f16_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (!layer || !layer->blob || !x || !y || M < 1) return PBL_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(layer->blob) & 15) return PBL_ERR_MISALIGNED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t esz = y_f32 ? 4 : 2;
    const Route r = route_of(layer, M, !(reinterpret_cast<uintptr_t>(x) & 15));
    if (r.mfma) {
        // Without the K-split workspace the matrix-core kernel leaves most of the chip idle on all but the largest layers:
        // a few tokens are then better served by GEMV passes.
        const size_t need = pbl_mfma_workspace_bytes(layer, M < 32 ? M : 32);
        const bool ws_ok = !need || (workspace && workspace_bytes >= need);
        if (ws_ok || M > 8) {
            const int rc = mfma_passes(layer, x, y, M, y_f32, workspace, workspace_bytes, stream);
            if (rc != PBL_ERR_UNSUPPORTED) return rc;   // unsupported (LDS budget): every pass failed the same way, fall through
        }
    }
    if (layer->G != 1) return linear_groups(layer, x, y, M, y_f32, st);
    const bool sf = layer->flags & PBL_FLAG_SAL_F16;
    const int split = r.split, wpb = r.wpb, mb_max = r.mb_max;
    for (int m0 = 0; m0 < M; m0 += mb_max) {
        const int mb = M - m0 < mb_max ? M - m0 : mb_max;
        GemvArgs a{};
        a.layer = *layer;
        a.x = static_cast<const _Float16*>(x) + size_t(m0) * layer->K;
        a.y = static_cast<char*>(y) + size_t(m0) * layer->N * esz;
        a.M = mb; a.y_f32 = y_f32; a.grouped = 0;
        const dim3 grid(split > 1 ? layer->NRB : (layer->NRB + wpb - 1) / wpb, 1, 1);
        const size_t lds = lds_bytes(layer->P, layer->max_nch, mb, wpb, split);
        int rc;
        switch (split) {
            case 8: rc = launch_split<8>(mb, sf, a, grid, lds, st); break;
            case 4: rc = launch_split<4>(mb, sf, a, grid, lds, st); break;
            case 2: rc = launch_split<2>(mb, sf, a, grid, lds, st); break;
            default: rc = wpb == 4 ? launch_mb<4>(mb, sf, a, grid, lds, st) : launch_mb<1>(mb, sf, a, grid, lds, st);
        }
        if (rc != PBL_OK) return rc;
    }
    return PBL_OK;
}

// bf16 activations in ONE launch (round 5): x [M, K] bf16, y [M, N] bf16 (fp32 with y_f32) = F.linear(x, W, bias) as the reference's
// bf16 runs compute it (qat/run_qat.py:120; quant/outlier_quantizer.py:101-106 under bf16) -- the GEMV's staging phase scales every
// token's row by a power of two into fp16's range while it copies it into LDS (exact; a token holding inf / NaN runs as its indicator
// row), the epilogue scales back, adds the bias and rounds to bf16: the arithmetic of pbl_act_bf16_prepare + pbl_linear_f16 +
// pbl_act_finish, bit for bit, without the two extra launches.  One GEMV pass of a group-free layer only (M <= the route's limit,
// at most 4): PBL_ERR_UNSUPPORTED otherwise -- the caller then runs the three-launch form, which serves every kernel family.
int pbl_linear_bf16(const pbl_layer* layer, const void* x_bf16, void* y, int M, int y_f32, void* stream) {
    if (!layer || !layer->blob || !x_bf16 || !y || M < 1) return PBL_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(layer->blob) & 15) return PBL_ERR_MISALIGNED;
    if (layer->G != 1) return PBL_ERR_UNSUPPORTED;
    const Route r = route_of(layer, M, true);
    if (M > r.mb_max) return PBL_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool sf = layer->flags & PBL_FLAG_SAL_F16;
    GemvArgs a{};
    a.layer = *layer;
    a.x = static_cast<const _Float16*>(x_bf16);                  // (bf16 bits: the XB instantiation reads them as such)
    a.y = y;
    a.M = M; a.y_f32 = y_f32; a.grouped = 0; a.x_bf16 = 1;
    const int split = r.split, wpb = r.wpb;
    const dim3 grid(split > 1 ? layer->NRB : (layer->NRB + wpb - 1) / wpb, 1, 1);
    const size_t lds = lds_bytes(layer->P, layer->max_nch, M, wpb, split);
    switch (split) {
        case 8: return launch_split<8>(M, sf, a, grid, lds, st);
        case 4: return launch_split<4>(M, sf, a, grid, lds, st);
        case 2: return launch_split<2>(M, sf, a, grid, lds, st);
        default: return wpb == 4 ? launch_mb<4>(M, sf, a, grid, lds, st) : launch_mb<1>(M, sf, a, grid, lds, st);
    }
}

// K-split tensor parallelism, fused: the GEMV of this rank's column shard with its epilogue pushing the fp32 partial y[M, N]
// straight into slot [set][rank] of every rank's peer-mapped communication buffer (pbl_comm_alloc / pbl_ipc_open; layout in
// csrc/pbl_p2p_layout.h) -- 7 xGMI stores + 1 local per row owner -- and counting every finished record at the peers.
// pbl_p2p_reduce_f32_dev(.., expect_records = layer->NRB, ..) then sums the slots: the partial never makes the round trip
// through local HBM and the push has no launch of its own.  One GEMV pass only (M <= 4 with the x tile in LDS, group-free
// layers): PBL_ERR_UNSUPPORTED otherwise -- the caller then runs pbl_linear_f16 + pbl_p2p_allreduce_f32_dev.
int pbl_linear_f16_push(const pbl_layer* layer, const void* x, int M, void* const* peer_bufs, int rank, int world, size_t max_elems,
                        void* stream) {
    if (!layer || !layer->blob || !x || !peer_bufs || M < 1 || world < 1 || world > PBL_P2P_MAX_WORLD || rank < 0 || rank >= world)
        return PBL_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(layer->blob) & 15) return PBL_ERR_MISALIGNED;
    const size_t cap = (max_elems + 3) & ~size_t(3);
    if (size_t(M) * layer->N > cap) return PBL_ERR_CAPACITY;
    if (layer->G != 1) return PBL_ERR_UNSUPPORTED;
    const Route r = route_of(layer, M, !(reinterpret_cast<uintptr_t>(x) & 15));
    if (M > r.mb_max) return PBL_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool sf = layer->flags & PBL_FLAG_SAL_F16;
    GemvArgs a{};
    a.layer = *layer;
    a.layer.bias = rank == 0 ? layer->bias : nullptr;         // (the bias is added once: by rank 0's partial)
    a.x = static_cast<const _Float16*>(x);
    a.y = nullptr;
    a.M = M; a.y_f32 = 1; a.grouped = 0;
    a.push_world = world; a.push_rank = rank; a.push_cap = cap;
    for (int p = 0; p < world; ++p) {
        if (!peer_bufs[p]) return PBL_ERR_INVALID_ARG;
        a.push_peer[p] = static_cast<uint8_t*>(peer_bufs[p]);
    }
    const int split = r.split, wpb = r.wpb;
    const dim3 grid(split > 1 ? layer->NRB : (layer->NRB + wpb - 1) / wpb, 1, 1);
    const size_t lds = lds_bytes(layer->P, layer->max_nch, M, wpb, split);
    switch (split) {
        case 8: return launch_split<8>(M, sf, a, grid, lds, st);
        case 4: return launch_split<4>(M, sf, a, grid, lds, st);
        case 2: return launch_split<2>(M, sf, a, grid, lds, st);
        default: return wpb == 4 ? launch_mb<4>(M, sf, a, grid, lds, st) : launch_mb<1>(M, sf, a, grid, lds, st);
    }
}

// The largest M pbl_linear_f16_push takes for this shard: route_of's one-pass limit, which follows the shard's fullest record
// (LDS: x tile + chunk partials).  The ranks of a K-split layer agree on the minimum (pb_llm_amd/parallel.py).
int pbl_linear_push_max_tokens(const pbl_layer* layer) {
    if (!layer || layer->G != 1) return 0;
    return route_of(layer, PBL_MAX_TOKENS_PER_LAUNCH, true).mb_max;
}

int pbl_unpack_dev(const pbl_layer* layer, void* W_out, int out_f32, void* stream) {
    if (!layer || !layer->blob || !W_out) return PBL_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(layer->blob) & 15) return PBL_ERR_MISALIGNED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    pbl_layer L = *layer;
    // one LDS row-segment buffer per wave: at most 16 KiB each, so that several workgroups share a CU
    const size_t esz = out_f32 ? 4 : 2;
    int seg_panels = int((16 * 1024) / (PBL_PANEL_COLS * esz));           // 16 (fp16) or 8 (fp32) panels
    if (seg_panels > int(layer->P)) seg_panels = int(layer->P);
    const size_t lds = size_t(seg_panels) * PBL_PANEL_COLS * esz * 4;
    void* argv[] = {&L, &W_out, &seg_panels};
    const void* k = out_f32 ? reinterpret_cast<const void*>(pbl_unpack_kernel<float>)
                            : reinterpret_cast<const void*>(pbl_unpack_kernel<_Float16>);
    return hipLaunchKernel(k, dim3(4 * layer->NRB), dim3(4 * PBL_WAVE), argv, lds, st) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

int pbl_gemv_f16_grouped(const pbl_layer* layers_dev, const void* const* x_dev, void* const* y_dev, int Lc,
                         int M, uint32_t max_NRB, uint32_t max_K, uint32_t max_nch, uint32_t max_nexc,
                         int any_groups, int y_f32, void* stream) {
    (void)max_nexc;
    if (!layers_dev || !x_dev || !y_dev || Lc < 1 || M < 1 || M > PBL_MAX_TOKENS_PER_LAUNCH)
        return PBL_ERR_INVALID_ARG;
    if (Lc > 65535) return PBL_ERR_UNSUPPORTED;
    GemvArgs a{};
    a.layers = layers_dev; a.xs = x_dev; a.ys = y_dev; a.M = M; a.y_f32 = y_f32; a.grouped = 1; a.Lc = Lc;
    if (any_groups & 1) return grouped_groups(a, M, max_NRB, max_K, max_nch, static_cast<hipStream_t>(stream));
#ifndef PBL_GROUPED_WPB
#define PBL_GROUPED_WPB 4
#endif
    const uint32_t P = (max_K + PBL_PANEL_COLS - 1) / PBL_PANEL_COLS;
    const bool sf = (any_groups & 2) != 0;   // bit 1: the group may contain PBL_FLAG_SAL_F16 layers
    hipStream_t st = static_cast<hipStream_t>(stream);
    // few records in total (fused q/k/v, gate+up at decode time): latency mode, S waves per record
    const int split = uint64_t(max_NRB) * uint64_t(Lc) >= PBL_SPLIT_TARGET_WAVES ? 1 : pick_split(max_NRB * uint32_t(Lc), P);
    if (split > 1) {
        const dim3 grid(max_NRB, Lc, 1);
        const size_t lds = lds_bytes(P, max_nch, M, split, split);
        switch (split) {
            case 8: return launch_split<8>(M, sf, a, grid, lds, st);
            case 4: return launch_split<4>(M, sf, a, grid, lds, st);
            default: return launch_split<2>(M, sf, a, grid, lds, st);
        }
    }
    const int wpb = PBL_GROUPED_WPB;
    const dim3 grid((max_NRB + wpb - 1) / wpb, Lc, 1);
    return launch_mb<PBL_GROUPED_WPB>(M, sf, a, grid, lds_bytes(P, max_nch, M, wpb), st);
}

static int fused_launch(GemvArgs& a, int Lc, int M, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, void* stream);

int pbl_gemv_f16_fused(const pbl_layer* layers_dev, const uint64_t* y_off_dev, const void* x, void* y, int Lc, int M,
                       uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, void* stream) {
    if (!layers_dev || !y_off_dev || !x || !y || Lc < 1 || M < 1 || M > PBL_MAX_TOKENS_PER_LAUNCH || !ldy)
        return PBL_ERR_INVALID_ARG;
    if (Lc > 65535) return PBL_ERR_UNSUPPORTED;
    GemvArgs a{};
    a.layers = layers_dev; a.M = M; a.y_f32 = y_f32; a.grouped = 1;
    a.x_shared = static_cast<const _Float16*>(x); a.y_shared = y; a.y_off = y_off_dev; a.ldy = ldy;
    return fused_launch(a, Lc, M, max_NRB, K, max_nch, group_flags, stream);
}

static int fused_host(const pbl_layer* layers_host, const uint64_t* y_off_host, const void* x, void* y, int Lc, int M,
                      uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, int x_bf16, void* stream);

int pbl_gemv_f16_fused_host(const pbl_layer* layers_host, const uint64_t* y_off_host, const void* x, void* y, int Lc, int M,
                            uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, void* stream) {
    return fused_host(layers_host, y_off_host, x, y, Lc, M, ldy, max_NRB, K, max_nch, group_flags, y_f32, 0, stream);
}

// the same launch with bf16 activations converted in the kernel and a bf16 (or fp32) result: see pbl_linear_bf16.  Group-free members
// only (PBL_ERR_UNSUPPORTED when bit 0 of group_flags is set: the column-group kernel has no bf16 instantiation).
int pbl_gemv_bf16_fused_host(const pbl_layer* layers_host, const uint64_t* y_off_host, const void* x_bf16, void* y, int Lc, int M,
                             uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, void* stream) {
    if (group_flags & 1) return PBL_ERR_UNSUPPORTED;
    return fused_host(layers_host, y_off_host, x_bf16, y, Lc, M, ldy, max_NRB, K, max_nch, group_flags, y_f32, 1, stream);
}

static int fused_host(const pbl_layer* layers_host, const uint64_t* y_off_host, const void* x, void* y, int Lc, int M,
                      uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, int x_bf16, void* stream) {
    if (!layers_host || !y_off_host || !x || !y || Lc < 1 || Lc > PBL_FUSED_INLINE_MAX || M < 1 || M > PBL_MAX_TOKENS_PER_LAUNCH || !ldy)
        return PBL_ERR_INVALID_ARG;
    GemvArgs a{};
    a.M = M; a.y_f32 = y_f32; a.grouped = 1; a.n_inl = Lc; a.x_bf16 = x_bf16;
    for (int l = 0; l < Lc; ++l) {
        if (!layers_host[l].blob || (reinterpret_cast<uintptr_t>(layers_host[l].blob) & 15)) return PBL_ERR_MISALIGNED;
        a.inl[l] = layers_host[l]; a.inl_off[l] = y_off_host[l];
    }
    a.x_shared = static_cast<const _Float16*>(x); a.y_shared = y; a.ldy = ldy;
    return fused_launch(a, Lc, M, max_NRB, K, max_nch, group_flags, stream);
}

static int fused_launch(GemvArgs& a, int Lc, int M, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, void* stream) {
    const int y_f32 = a.y_f32;
    (void)y_f32;
    if (group_flags & 1) { a.Lc = Lc; return grouped_groups(a, M, max_NRB, K, max_nch, static_cast<hipStream_t>(stream)); }
    const uint32_t P = (K + PBL_PANEL_COLS - 1) / PBL_PANEL_COLS;
    const bool sf = (group_flags & 2) != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int split = uint64_t(max_NRB) * uint64_t(Lc) >= PBL_SPLIT_TARGET_WAVES ? 1 : pick_split(max_NRB * uint32_t(Lc), P);
    if (split > 1) {
        const dim3 grid(max_NRB, Lc, 1);
        const size_t lds = lds_bytes(P, max_nch, M, split, split);
        switch (split) {
            case 8: return launch_split<8>(M, sf, a, grid, lds, st);
            case 4: return launch_split<4>(M, sf, a, grid, lds, st);
            default: return launch_split<2>(M, sf, a, grid, lds, st);
        }
    }
    const int wpb = PBL_GROUPED_WPB;
    const dim3 grid((max_NRB + wpb - 1) / wpb, Lc, 1);
    return launch_mb<PBL_GROUPED_WPB>(M, sf, a, grid, lds_bytes(P, max_nch, M, wpb), st);
}

#ifdef PBL_PROTO_ROWMAJOR
#define PBL_PROTO_SECTION_HOST
#include "../../tools/proto_rowmajor_kernel.inc"
#undef PBL_PROTO_SECTION_HOST
#endif

}  // extern "C"
