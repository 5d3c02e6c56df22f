// ubench_xcd_reduce.hip -- can K-split partial tiles be summed INSIDE one launch through an XCD's own L2?
//
// Round 4 folded the small-batch kernel's K-split sum into the last split to arrive and lost: with agent-scope release / acquire
// fences (a buffer_wbl2 per wave) 44 - 110 us per launch, with agent-scope stores / loads 36 - 61 us, against 25 us with a second
// launch (csrc/pbl_gemm_img.hip).  An MI355X has eight XCDs, each with its own L2 that is coherent for the CUs of THAT XCD.  If all
// K splits of an output tile are computed on one XCD, the exchange needs no agent-scope traffic at all: plain stores reach the XCD's
// L2 (the vector L1 is write-through), an atomic WITHOUT the sc1 bit executes in that L2, and the reader only has to drop its own L1
// (buffer_inv sc0) -- the "threadgroup split" rules of the gfx942 memory model.  This program tests exactly that mechanism, outside
// the product, with the work assigned by the XCC_ID a workgroup actually runs on:
//   * per-XCD ticket queues: a workgroup reads HW_REG_XCC_ID, draws tickets from ITS XCD's queue (L2-local atomic) and processes
//     items (column, split) of columns with column % 8 == xcd until the queue is empty -- whatever the dispatcher's placement is,
//     all splits of a column meet in one L2;
//   * per-column arrival counter (L2-local atomic); the last arriver drops its L1 and sums the KS partial tiles in split order;
//   * two sets of queues / counters alternate by an epoch word: the globally last workgroup (agent-scope done counter) zeroes the
//     OTHER set for the next launch, which no other workgroup touches in this launch.
// It verifies every sum of every iteration against the host (new data each iteration: a stale L1 / L2 line would show), reports how
// many items each XCD processed, and times the launch against the two-launch form.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int T = 1024;                 // floats per partial tile (4 per thread)
constexpr int NQ = 8;                   // XCDs
constexpr int TSTRIDE = 32;             // words between the ticket counters of two XCDs (their own 128-byte lines)

struct Ctl {                            // device layout (words)
    uint32_t epoch, done, pad[30];
    // set s at words 32 + s * SETW: ticket[NQ * TSTRIDE], cnt[C], xcd_items[NQ] (statistics)
};

__device__ __forceinline__ float gen(uint32_t col, uint32_t ks, uint32_t i, uint32_t seed) {
    uint32_t h = (col * 2654435761u) ^ (ks * 40503u + 17u) ^ (i * 2246822519u) ^ (seed * 3266489917u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return float(int(h & 0xFFFF) - 32768) * (1.0f / 256.0f);
}

__global__ __launch_bounds__(256) void fused_kernel(uint32_t* ctl, float* part, float* out, int C, int KS, uint32_t seed, uint32_t setw,
                                                    uint32_t* stats) {
    __shared__ uint32_t s_t, s_old;
    const int tid = threadIdx.x;
    const uint32_t xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;         // HW_REG_XCC_ID, bits [3:0]
    const uint32_t e = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t* set = ctl + 32 + (e & 1u) * setw;
    uint32_t* ticket = set + xcd * TSTRIDE;
    uint32_t* cnt = set + NQ * TSTRIDE;
    const uint32_t ncol_x = (uint32_t(C) > xcd) ? (uint32_t(C) - xcd + 7u) / 8u : 0u;
    const uint32_t qlen = ncol_x * uint32_t(KS);
    uint32_t mine = 0;
    for (;;) {
        if (tid == 0) s_t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (no sc1: executes in THIS XCD's L2)
        __syncthreads();
        const uint32_t t = s_t;
        if (t >= qlen) break;
        const uint32_t col = xcd + 8u * (t / uint32_t(KS)), ks = t % uint32_t(KS);
        float4 v;
        v.x = gen(col, ks, 4 * tid, seed); v.y = gen(col, ks, 4 * tid + 1, seed); v.z = gen(col, ks, 4 * tid + 2, seed); v.w = gen(col, ks, 4 * tid + 3, seed);
        reinterpret_cast<float4*>(part + (size_t(ks) * C + col) * T)[tid] = v;                                        // plain store: write-through to L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_old = __hip_atomic_fetch_add(cnt + col, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        if (s_old == uint32_t(KS) - 1u) {                                       // the last split of this column to arrive: sum in split order
            asm volatile("buffer_inv sc0" ::: "memory");                         // drop this CU's vector L1: the other splits' tiles are in L2
            float4 s = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < KS; ++k) {
                const float4 p = reinterpret_cast<const float4*>(part + (size_t(k) * C + col) * T)[tid];
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
            reinterpret_cast<float4*>(out + size_t(col) * T)[tid] = s;
            if (tid == 0) cnt[col] = 0u;
        }
        ++mine;
        __syncthreads();
    }
    if (tid == 0) {
        if (mine) atomicAdd(stats + xcd, mine);
        const uint32_t total = gridDim.x;
        // (relaxed: an agent-scope release would write this XCD's whole L2 back, once per workgroup)
        const uint32_t d = __hip_atomic_fetch_add(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == total - 1u) {                                                   // globally last: prepare the OTHER set for the next launch
            uint32_t* other = ctl + 32 + ((e & 1u) ^ 1u) * setw;
            for (int q = 0; q < NQ; ++q) other[q * TSTRIDE] = 0u;
            __hip_atomic_store(ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl, e + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the two-launch form: every workgroup writes its partial tile; a second kernel sums
__global__ __launch_bounds__(256) void split_kernel(float* part, int C, int KS, uint32_t seed) {
    const int tid = threadIdx.x;
    const uint32_t col = blockIdx.x % uint32_t(C), ks = blockIdx.x / uint32_t(C);
    float4 v;
    v.x = gen(col, ks, 4 * tid, seed); v.y = gen(col, ks, 4 * tid + 1, seed); v.z = gen(col, ks, 4 * tid + 2, seed); v.w = gen(col, ks, 4 * tid + 3, seed);
    reinterpret_cast<float4*>(part + (size_t(ks) * C + col) * T)[tid] = v;
}
__global__ __launch_bounds__(256) void reduce_kernel(const float* part, float* out, int C, int KS) {
    const int tid = threadIdx.x;
    const uint32_t col = blockIdx.x;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < KS; ++k) {
        const float4 p = reinterpret_cast<const float4*>(part + (size_t(k) * C + col) * T)[tid];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    reinterpret_cast<float4*>(out + size_t(col) * T)[tid] = s;
}

static float hgen(uint32_t col, uint32_t ks, uint32_t i, uint32_t seed) {
    uint32_t h = (col * 2654435761u) ^ (ks * 40503u + 17u) ^ (i * 2246822519u) ^ (seed * 3266489917u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return float(int(h & 0xFFFF) - 32768) * (1.0f / 256.0f);
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 108, KS = argc > 2 ? atoi(argv[2]) : 4, iters = argc > 3 ? atoi(argv[3]) : 300;
    const uint32_t setw = NQ * TSTRIDE + uint32_t((C + 31) & ~31);
    const size_t ctl_words = 32 + 2 * size_t(setw);
    uint32_t *ctl, *stats;
    float *part, *out, *out2;
    CK(hipMalloc(&ctl, ctl_words * 4)); CK(hipMemset(ctl, 0, ctl_words * 4));
    CK(hipMalloc(&stats, 64)); CK(hipMemset(stats, 0, 64));
    CK(hipMalloc(&part, size_t(KS) * C * T * 4)); CK(hipMalloc(&out, size_t(C) * T * 4)); CK(hipMalloc(&out2, size_t(C) * T * 4));
    std::vector<float> h(size_t(C) * T), ref(size_t(C) * T);
    int bad_iters = 0;
    long bad_vals = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t seed = 1000u + uint32_t(it);
        CK(hipMemsetAsync(out, 0xFF, size_t(C) * T * 4));                       // NaN: an unprocessed column shows
        fused_kernel<<<C * KS, 256>>>(ctl, part, out, C, KS, seed, setw, stats);
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < T; ++i) {
                float s = 0.f;
                for (int k = 0; k < KS; ++k) s += hgen(c, k, i, seed);
                if (!(h[size_t(c) * T + i] == s)) ++bad;
            }
        if (bad) { ++bad_iters; bad_vals += bad; if (bad_iters <= 3) printf("iteration %d: %ld wrong values\n", it, bad); }
    }
    uint32_t hs[16];
    CK(hipMemcpy(hs, stats, 64, hipMemcpyDeviceToHost));
    printf("C=%d KS=%d iterations=%d: %d iterations with wrong sums (%ld values); items per XCD over all iterations:", C, KS, iters, bad_iters, bad_vals);
    for (int q = 0; q < NQ; ++q) printf(" %u", hs[q]);
    printf("\n");
    // timing: fused vs two launches, 500 launches each
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 500; ++i) fused_kernel<<<C * KS, 256>>>(ctl, part, out, C, KS, 7u + i, setw, stats);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const float f_us = ms * 2.f;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 500; ++i) { split_kernel<<<C * KS, 256>>>(part, C, KS, 7u + i); reduce_kernel<<<C, 256>>>(part, out2, C, KS); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pass %d: fused %.2f us per launch, two launches %.2f us\n", rep, f_us, ms * 2.f);
    }
    printf("%s\n", bad_iters ? "FAIL" : "PASS");
    return bad_iters ? 1 : 0;
}
