// Micro-benchmark: per-wave issue cost (shader cycles) of the VALU / LDS instructions the
// PB GEMV kernel is built from, on gfx950.  hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define REP 64
#define ITER 256

template <int OP>
__global__ void bench(uint32_t* out, uint64_t* cycles, uint32_t seed) {
    __shared__ _Float16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = _Float16(i & 15);
    __syncthreads();
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    float f0 = 0, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    const uint32_t m = 0x80008000u, c = 0x3C003C00u;
    uint32_t cv = c + (seed & 1);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) {  // v_and_b32 (literal mask)
                asm volatile("v_and_b32 %0, 0x80008000, %0\n v_and_b32 %1, 0x80008000, %1\n v_and_b32 %2, 0x80008000, %2\n v_and_b32 %3, 0x80008000, %3\n"
                             "v_and_b32 %4, 0x80008000, %4\n v_and_b32 %5, 0x80008000, %5\n v_and_b32 %6, 0x80008000, %6\n v_and_b32 %7, 0x80008000, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 1) {  // v_and_or_b32 (sgpr mask, vgpr const)
                asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n"
                             "v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(cv));
            } else if (OP == 2) {  // v_dot2c_f32_f16
                asm volatile("v_dot2c_f32_f16 %0, %8, %9\n v_dot2c_f32_f16 %1, %8, %9\n v_dot2c_f32_f16 %2, %8, %9\n v_dot2c_f32_f16 %3, %8, %9\n"
                             "v_dot2c_f32_f16 %4, %8, %9\n v_dot2c_f32_f16 %5, %8, %9\n v_dot2c_f32_f16 %6, %8, %9\n v_dot2c_f32_f16 %7, %8, %9\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a0), "v"(cv));
            } else if (OP == 3) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                             "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a0), "v"(cv));
            } else if (OP == 4) {  // v_fma_mix_f32
                asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel_hi:[0,1,0]\n"
                             "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel_hi:[0,1,0]\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a0), "v"(cv));
            } else if (OP == 5) {  // v_pk_fma_f16
                asm volatile("v_pk_fma_f16 %0, %8, %9, %0\n v_pk_fma_f16 %1, %8, %9, %1\n v_pk_fma_f16 %2, %8, %9, %2\n v_pk_fma_f16 %3, %8, %9, %3\n"
                             "v_pk_fma_f16 %4, %8, %9, %4\n v_pk_fma_f16 %5, %8, %9, %5\n v_pk_fma_f16 %6, %8, %9, %6\n v_pk_fma_f16 %7, %8, %9, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cv), "v"(cv));
            } else if (OP == 6) {  // v_cvt_f32_ubyte0
                asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                             "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9\n"
                             : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3), "=v"(f4), "=v"(f5), "=v"(f6), "=v"(f7) : "v"(a0), "v"(cv));
            } else if (OP == 7) {  // v_add_u32 sdwa byte
                asm volatile("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
                             "v_add_u32_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_u32_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                             "v_add_u32_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cv), "v"(seed));
            } else if (OP == 8) {  // ds_read_u16 random-ish addresses, 8 in flight
                uint32_t b0 = (a0 & 0x1FFE), b1 = (a1 & 0x1FFE), b2 = (a2 & 0x1FFE), b3 = (a3 & 0x1FFE), b4 = (a4 & 0x1FFE), b5 = (a5 & 0x1FFE), b6 = (a6 & 0x1FFE), b7 = (a7 & 0x1FFE);
                uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
                asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %9\n ds_read_u16 %2, %10\n ds_read_u16 %3, %11\n ds_read_u16 %4, %12\n ds_read_u16 %5, %13\n ds_read_u16 %6, %14\n ds_read_u16 %7, %15\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                             : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
                a0 += r0 * 2 + 2; a1 += r1 * 2 + 6; a2 += r2 * 2 + 10; a3 += r3 * 2 + 14; a4 += r4 * 2 + 18; a5 += r5 * 2 + 22; a6 += r6 * 2 + 26; a7 += r7 * 2 + 30;
            } else if (OP == 9) {  // v_perm_b32
                asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                             "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cv), "s"(0x07000601u));
            } else if (OP == 10) {  // v_lshl_add_u32
                asm volatile("v_lshl_add_u32 %0, %8, 1, %0\n v_lshl_add_u32 %1, %8, 1, %1\n v_lshl_add_u32 %2, %8, 1, %2\n v_lshl_add_u32 %3, %8, 1, %3\n"
                             "v_lshl_add_u32 %4, %8, 1, %4\n v_lshl_add_u32 %5, %8, 1, %5\n v_lshl_add_u32 %6, %8, 1, %6\n v_lshl_add_u32 %7, %8, 1, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cv));
            }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256, threads = 256 * waves_per_simd;  // 1 block per CU
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, sizeof(uint32_t) * blocks * threads);
    hipMalloc(&cyc, sizeof(uint64_t) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<OP><<<blocks, threads>>>(out, cyc, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<OP><<<blocks, threads>>>(out, cyc, 2);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(uint64_t) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += double(v); avg /= blocks;
    const double n = double(ITER) * REP;
    // per-SIMD instruction issue time in ns: wall / (instructions per SIMD)
    const double ns_per_inst_simd = ms * 1e6 / (n * waves_per_simd);
    printf("%-22s waves/SIMD=%d  memtime-ticks/inst(one wave)=%.3f  wall ns per wave-inst per SIMD=%.3f  (%.2f cyc @2.4GHz)\n",
           name, waves_per_simd, avg / n, ns_per_inst_simd, ns_per_inst_simd * 2.4);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("v_and_b32(lit)", w);
        run<1>("v_and_or_b32", w);
        run<2>("v_dot2c_f32_f16", w);
        run<3>("v_fma_f32", w);
        run<4>("v_fma_mix_f32", w);
        run<5>("v_pk_fma_f16", w);
        run<6>("v_cvt_f32_ubyteN", w);
        run<7>("v_add_u32_sdwa", w);
        run<8>("ds_read_u16 x8+wait", w);
        run<9>("v_perm_b32", w);
        run<10>("v_lshl_add_u32", w);
    }
    return 0;
}
