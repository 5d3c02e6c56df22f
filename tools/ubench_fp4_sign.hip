// Feasibility study for a NEXT-format sign plane (not part of libpbl): the GEMV's sign-plane phase on the MX matrix-core
// path of gfx950 instead of the VALU.
//
//   today        lane-owns-columns plane; per 2 weights one v_and_or (bit -> two-valued fp16) + one v_dot2_f32_f16:
//                34 VALU per 32 weights and lane, 1088 per record (16 rows x 4096 columns)
//   studied      ROW-MAJOR plane: lane (row = l & 15, kb = l >> 4) holds 32 sign bits of ITS row; one v_and_or per 8 weights
//                turns bit j of every nibble into a two-valued FP4 (E2M1) code ("nibble classes": bit 0 -> {1, 1.5},
//                bit 1 -> {0, 1}, bit 2 -> {0, 2}, bit 3 -> {1, -1}); v_mfma_scale_f32_16x16x128_f8f6f4 (A = fp4, B = fp8)
//                contracts 16 rows x 128 columns; x rides as three fp8 terms (x = t1 + t2 + t3 to fp16 accuracy) in three of the
//                sixteen token columns, pre-scaled by the class factor of its column; the per-class offsets are one dot
//                product with x per launch.
//
// Part 1 finds the operand layout of the scaled MFMA by comparing against a host model (the guides defer to an ISA document
// that is not in this image).  Part 2 times both formulations of the sign phase on resident data (compute only): cycles per
// 2048 weights and wave, one to four waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench_fp4 tools/ubench_fp4_sign.hip && ./ubench_fp4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static const float FP4[16] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f, -0.f, -.5f, -1.f, -1.5f, -2.f, -3.f, -4.f, -6.f};

static float fp8_e4m3(uint8_t b) {          // OCP E4M3 (bias 7, no infinities)
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e ? std::ldexp(1.f + m / 8.f, e - 7) : std::ldexp(m / 8.f, -6);
    return s ? -v : v;
}

// ---- part 1: one MFMA with caller-built operands ----------------------------------------------------------------------
template <int FA, int FB>
__global__ void one_mfma(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, float* __restrict__ d) {
    const int l = threadIdx.x;
    v8i A, B;
    for (int q = 0; q < 8; ++q) { A[q] = int(a[l * 8 + q]); B[q] = int(b[l * 8 + q]); }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, FA, FB, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

// fragments from sign bits: one dword of row-major bits per lane -> four v_and_or -> the FP4 operand (as the prototype kernel does)
__global__ void bits_mfma(const uint32_t* __restrict__ wbits, const uint32_t* __restrict__ b, float* __restrict__ d, const uint32_t* __restrict__ sc) {
    const int l = threadIdx.x;
    const uint32_t w = wbits[l];
    uint32_t c22 = 0x22222222u, c00 = 0u;
    asm volatile("" : "+v"(c22), "+v"(c00));
    uint32_t f0, f1, f2, f3;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f0) : "v"(w), "s"(0x11111111u), "v"(c22));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f1) : "v"(w), "s"(0x22222222u), "v"(c00));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f2) : "v"(w), "s"(0x44444444u), "v"(c00));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f3) : "v"(w), "s"(0x88888888u), "v"(c22));
    const v8i A = {int(f0), int(f1), int(f2), int(f3), 0, 0, 0, 0};
    v8i B;
    for (int q = 0; q < 8; ++q) B[q] = int(b[l * 8 + q]);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 4, 0, 0, 0x7f7f7f7f, 0, int(sc[l]));
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

// probe: which fp4 element of the A operand meets column k0 of an fp8 B (layout validated by the fp8 x fp8 test)?
// pattern 0: every fp4 element holds code (e % 8) -> values 0 .5 1 1.5 2 3 4 6; pattern 1: code (e / 8); pattern 2: code kb
__global__ void probe(int pattern, float* __restrict__ out) {
    const int l = threadIdx.x, kb = l >> 4;
    v8i A = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 32; ++e) {
        const uint32_t code = pattern == 0 ? (e % 8) : (pattern == 1 ? (e / 8) : kb);
        A[e / 8] |= int(code << (4 * (e % 8)));
    }
    for (int k0 = 0; k0 < 128; ++k0) {
        v8i B = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((l & 15) == 0 && kb == k0 / 32) { const int e = k0 % 32; B[e / 4] = int(0x38u << (8 * (e % 4))); }
        v4f c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 4, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        if ((l & 15) == 0) for (int r = 0; r < 4; ++r) out[k0 * 16 + 4 * kb + r] = c[r];
    }
}

// ---- part 2: the sign phase, compute only -----------------------------------------------------------------------------
#define KSTEPS 32          // 32 x 128 columns = one 4096-column record
template <int MODE>
__global__ __launch_bounds__(256) void sign_phase(float* __restrict__ out, uint64_t* __restrict__ cyc, uint32_t seed, int reps) {
    __shared__ __attribute__((aligned(16))) uint8_t xs[3 * 4096 + 512];     // three fp8 terms of x (MODE 1) / fp16 x (MODE 0: 8 KB)
    for (int i = threadIdx.x; i < int(sizeof(xs)); i += blockDim.x) xs[i] = uint8_t(0x30 + ((i * 7 + seed) & 15));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t w = seed * 2654435761u + threadIdx.x * 40503u;
    float total = 0.f;
    const uint64_t t0 = __builtin_readcyclecounter();
    if (MODE == 1) {
        const int tok = lane & 15, kb = lane >> 4;
        uint32_t c22 = 0x22222222u, c00 = 0u;
        asm volatile("" : "+v"(c22), "+v"(c00));
        for (int rep = 0; rep < reps; ++rep) {
            v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int ks = 0; ks < KSTEPS; ++ks) {
                w = w * 1664525u + 1013904223u;               // stands for the next dword of the lane's row (a register in the real kernel)
                v8i A = {0, 0, 0, 0, 0, 0, 0, 0};
                // nibble classes: bit 0 -> {1, 1.5} (| 0x2), bit 1 -> {0, 1}, bit 2 -> {0, 2}, bit 3 -> {1, -1} (| 0x2)
                uint32_t f0, f1, f2, f3;
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f0) : "v"(w), "s"(0x11111111u), "v"(c22));
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f1) : "v"(w), "s"(0x22222222u), "v"(c00));
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f2) : "v"(w), "s"(0x44444444u), "v"(c00));
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(f3) : "v"(w), "s"(0x88888888u), "v"(c22));
                A[0] = int(f0); A[1] = int(f1); A[2] = int(f2); A[3] = int(f3);
                // x: token columns 0..2 carry the three fp8 terms; the other columns read term 0 (their outputs are ignored)
                v8i B = {0, 0, 0, 0, 0, 0, 0, 0};
                if (tok < 3) {      // only the 12 lanes that carry real x terms touch the LDS
                    const uint8_t* xp = xs + tok * 4096 + 128 + ks * 128 + kb * 32 + tok * 32;
                    const u32x4 b0 = *reinterpret_cast<const u32x4*>(xp), b1 = *reinterpret_cast<const u32x4*>(xp + 16);
                    B = v8i{int(b0[0]), int(b0[1]), int(b0[2]), int(b0[3]), int(b1[0]), int(b1[1]), int(b1[2]), int(b1[3])};
                }
                acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 4, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
            total += acc[0] + acc[1] + acc[2] + acc[3];
        }
    } else {
        // today's formulation: per word (2 columns x 16 rows) 1 shift + 16 v_and(_or) + 17 v_dot2; 128 words = 2048 weights x 2 ...
        // one "k-step" here = the same 2048 weights per wave = 1 word per lane
        uint32_t c_one = 0x3C003C00u;
        asm volatile("" : "+v"(c_one));
        const uint32_t* xw = reinterpret_cast<const uint32_t*>(xs) + lane;
        for (int rep = 0; rep < reps; ++rep) {
            float acc[16], xl = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
            for (int ks = 0; ks < KSTEPS; ++ks) {
                w = w * 1664525u + 1013904223u;
                const uint32_t x2 = xw[(ks & 31) * 64];
                const uint32_t ws = w << 8;
                xl = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, c_one), __builtin_bit_cast(h2, x2), xl, false);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t m = 0x01000100u << c;
                    uint32_t t0_, t1_;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(t0_) : "v"(w), "s"(m), "v"(c_one));
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(t1_) : "v"(ws), "s"(m), "v"(c_one));
                    acc[c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, t0_), __builtin_bit_cast(h2, x2), acc[c], false);
                    acc[8 + c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, t1_), __builtin_bit_cast(h2, x2), acc[8 + c], false);
                }
            }
            for (int r = 0; r < 16; ++r) total += acc[r];
            total += xl;
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    // ---- part 1: layout ----------------------------------------------------------------------------------------------
    // hypotheses: A/B element e (0..31) of lane (idx = l & 15, kb = l >> 4) is k = kb * 32 + e [H0] or k = e * 4 + kb [H1];
    // fp4 elements fill a dword from the low nibble up, fp8 from the low byte up; D: col = l & 15, row = 4 (l >> 4) + reg
    std::vector<uint8_t> Ac(16 * 128), Bc(128 * 16);
    srand(7);
    for (auto& v : Ac) v = uint8_t(rand() & 15);
    const uint8_t b_vals[6] = {0x00, 0x38, 0xB8, 0x30, 0x40, 0xC0};      // 0, 1, -1, 0.5, 2, -2
    for (auto& v : Bc) v = b_vals[rand() % 6];
    // format combos: (A, B) in {fp8 x fp8, fp4 x fp4, fp4 x fp8}; element e of a lane: k = 32 kb + e, packed from the low
    // nibble / byte of register 0 upward
    for (int combo = 0; combo < 3; ++combo) {
        const bool a4 = combo != 0, b4 = combo == 1;
        std::vector<uint32_t> a(64 * 8, 0), b(64 * 8, 0);
        std::vector<float> Av(16 * 128), Bv(128 * 16);
        for (int l = 0; l < 64; ++l) {
            const int idx = l & 15, kb = l >> 4;
            for (int e = 0; e < 32; ++e) {
                // element e of lane group kb: k = 32 kb + e for an 8-bit operand; a 4-bit operand covers K in 16-element blocks
                // t = 4 (kb & 1) + 2 (e / 16) + (kb >> 1)  (found with the one-hot probe below; what matters for a kernel is only
                // that both operands of a product use the same k)
                const int k8 = kb * 32 + e, k4 = 16 * (4 * (kb & 1) + 2 * (e / 16) + (kb >> 1)) + e % 16;
                const int ka = a4 ? k4 : k8, kbb = b4 ? k4 : k8;
                const uint8_t ca = a4 ? Ac[idx * 128 + ka] : b_vals[Ac[idx * 128 + ka] % 6];
                const uint8_t cb = b4 ? uint8_t((Bc[kbb * 16 + idx] * 7 + kbb) & 15) : Bc[kbb * 16 + idx];
                Av[idx * 128 + ka] = a4 ? FP4[ca] : fp8_e4m3(ca);
                Bv[kbb * 16 + idx] = b4 ? FP4[cb] : fp8_e4m3(cb);
                if (a4) a[l * 8 + e / 8] |= uint32_t(ca) << (4 * (e % 8)); else a[l * 8 + e / 4] |= uint32_t(ca) << (8 * (e % 4));
                if (b4) b[l * 8 + e / 8] |= uint32_t(cb) << (4 * (e % 8)); else b[l * 8 + e / 4] |= uint32_t(cb) << (8 * (e % 4));
            }
        }
        uint32_t *da, *db; float* dd;
        CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dd, 256 * 4));
        CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        if (combo == 0) hipLaunchKernelGGL((one_mfma<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
        else if (combo == 1) hipLaunchKernelGGL((one_mfma<4, 4>), dim3(1), dim3(64), 0, 0, da, db, dd);
        else hipLaunchKernelGGL((one_mfma<4, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
        std::vector<float> d(256);
        CK(hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxerr_t = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int col = l & 15, row = 4 * (l >> 4) + r;
                double ref = 0, ref_t = 0;
                for (int k = 0; k < 128; ++k) { ref += double(Av[row * 128 + k]) * Bv[k * 16 + col]; ref_t += double(Av[col * 128 + k]) * Bv[k * 16 + row]; }
                maxerr = std::fmax(maxerr, std::fabs(ref - d[l * 4 + r]));
                maxerr_t = std::fmax(maxerr_t, std::fabs(ref_t - d[l * 4 + r]));
            }
        printf("{\"part\": \"layout\", \"formats\": \"%s\", \"max_abs_err\": %.4g, \"max_abs_err_if_D_transposed\": %.4g}\n",
               combo == 0 ? "fp8 x fp8" : (combo == 1 ? "fp4 x fp4" : "fp4 x fp8"), maxerr, maxerr_t);
        (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
    }
    {   // sign bits -> fragments -> MFMA against the host model of the prototype's layout
        std::vector<uint32_t> wb(64), b(64 * 8, 0);
        for (auto& v : wb) v = uint32_t(rand()) * 2654435761u + uint32_t(rand());
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 32; ++e) b[l * 8 + e / 4] |= uint32_t(Bc[((l >> 4) * 32 + e) * 16 + (l & 15)]) << (8 * (e % 4));
        for (int variant = 0; variant < 3; ++variant) {
        // scale of B's lane (col, kb): variant 0 all 1.0; 1: per-lane exponents (byte replicated); 2: the same, byte 0 only
        std::vector<uint32_t> sc(64);
        std::vector<int> sexp(64, 0);
        for (int l = 0; l < 64; ++l) { sexp[l] = variant ? (rand() % 7) - 3 : 0; const uint32_t by = uint32_t(127 + sexp[l]); sc[l] = variant == 2 ? by : by * 0x01010101u; }
        uint32_t *dw, *db, *ds; float* dd;
        CK(hipMalloc(&dw, 256)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dd, 1024)); CK(hipMalloc(&ds, 256));
        CK(hipMemcpy(dw, wb.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(ds, sc.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(bits_mfma, dim3(1), dim3(64), 0, 0, dw, db, dd, ds);
        std::vector<float> d(256);
        CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
        const float lo_v[4] = {1.f, 0.f, 0.f, 1.f}, hi_v[4] = {1.5f, 1.f, 2.f, -1.f};
        double maxerr = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int col = l & 15, row = 4 * (l >> 4) + r;
                double ref = 0;
                for (int kb = 0; kb < 4; ++kb)
                    for (int bit = 0; bit < 32; ++bit) {
                        const int n = bit / 4, c = bit % 4, e = 8 * c + n;
                        const int k = 16 * (4 * (kb & 1) + 2 * (e / 16) + (kb >> 1)) + e % 16;
                        const float v = ((wb[kb * 16 + row] >> bit) & 1) ? hi_v[c] : lo_v[c];
                        // which lane's scale applies to B's column-k element: the K block in the 4-bit operand's order (= kb here)
                        ref += double(v) * fp8_e4m3(Bc[k * 16 + col]) * std::ldexp(1.0, sexp[kb * 16 + col]);
                    }
                maxerr = std::fmax(maxerr, std::fabs(ref - d[l * 4 + r]));
            }
        printf("{\"part\": \"layout\", \"formats\": \"sign bits -> v_and_or fragments (fp4) x fp8, scale variant %d\", \"max_abs_err\": %.4g}\n", variant, maxerr);
        (void)hipFree(dw); (void)hipFree(db); (void)hipFree(dd); (void)hipFree(ds);
        }
    }
    {   // fp4 (A) against fp8 (B): decode the element of A that column k0 of B meets
        float* po; CK(hipMalloc(&po, 3 * 128 * 16 * 4));
        for (int p = 0; p < 3; ++p) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, po + p * 2048);
        std::vector<float> pr(3 * 2048);
        CK(hipMemcpy(pr.data(), po, pr.size() * 4, hipMemcpyDeviceToHost));
        auto code_of = [](float v) { for (int c = 0; c < 8; ++c) if (FP4[c] == v) return c; return -1; };
        printf("{\"part\": \"probe\", \"k0 -> (kb, e) of the fp4 operand (row 0)\": [");
        for (int k0 = 0; k0 < 128; ++k0) {
            const int e = code_of(pr[k0 * 16]) + 8 * code_of(pr[2048 + k0 * 16]), kb = code_of(pr[4096 + k0 * 16]);
            printf("%s[%d,%d]", k0 ? "," : "", kb, e);
        }
        printf("]}\n");
        (void)hipFree(po);
    }
    // ---- part 2: timing ----------------------------------------------------------------------------------------------
    float* out; uint64_t* cyc;
    const int max_blocks = 256 * 4;
    CK(hipMalloc(&out, size_t(max_blocks) * 256 * 4)); CK(hipMalloc(&cyc, max_blocks * 8));
    const int reps = 64;
    for (int mode = 0; mode < 2; ++mode)
        for (int wps = 1; wps <= 4; ++wps) {            // waves per SIMD: blocks of 4 waves, `wps` blocks per CU
            const int blocks = 256 * wps;
            for (int it = 0; it < 3; ++it) {
                if (mode == 0) hipLaunchKernelGGL(sign_phase<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, 123u + it, reps);
                else hipLaunchKernelGGL(sign_phase<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, 123u + it, reps);
            }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            const int n = 20;
            for (int it = 0; it < n; ++it) {
                if (mode == 0) hipLaunchKernelGGL(sign_phase<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, 5u + it, reps);
                else hipLaunchKernelGGL(sign_phase<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, 5u + it, reps);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // work per launch: blocks * 4 waves * reps records' sign phases; per SIMD: wps waves
            const double records = double(blocks) * 4 * reps;
            const double us_per_record_per_simd = double(ms) * 1e3 / n / (records / 1024.0);
            printf("{\"part\": \"timing\", \"formulation\": \"%s\", \"waves_per_simd\": %d, \"us_per_launch\": %.1f, "
                   "\"simd_us_per_record_sign_phase\": %.3f, \"cycles_at_2.4GHz\": %.0f}\n",
                   mode == 0 ? "valu (v_and_or + v_dot2, today)" : "mx fp4 x fp8 mfma (row-major plane)", wps, ms * 1e3 / n,
                   us_per_record_per_simd, us_per_record_per_simd * 2400.0);
        }
    return 0;
}
