// ubench_cu_bw.hip -- how much HBM bandwidth does ONE CU pull, and how does the chip's rate depend on the number of CUs that stream?
// (round 6: the small-batch kernel's whole-K geometry runs 216 workgroups, one per CU; is 216 / 256 of the chip's rate its ceiling?)
// Each workgroup streams its own contiguous chunk of a large buffer with `unroll` independent 16-byte loads per lane in flight.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_cu_bw.hip -o build/ubench_cu_bw ; run: build/ubench_cu_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ src, size_t per_wg_vec, unsigned* __restrict__ sink) {
    const u32x4* p = src + size_t(blockIdx.x) * per_wg_vec;
    u32x4 acc = {0, 0, 0, 0};
    const size_t step = size_t(blockDim.x) * U;
    for (size_t i = threadIdx.x; i + step <= per_wg_vec + threadIdx.x; i += step) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + size_t(u) * blockDim.x);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    const size_t bytes = size_t(2) << 30;                      // 2 GiB: every launch reads `total` bytes of it, far beyond the caches
    u32x4* buf;
    unsigned* sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[] = {32, 64, 128, 192, 216, 240, 256, 432, 512, 1024};
    const int threads[] = {256, 512, 1024};
    printf("{\"what\": \"streaming read, one contiguous chunk per workgroup, nontemporal 16-byte loads\", \"rows\": [\n");
    for (int t : threads)
        for (int g : grids) {
            const size_t total = size_t(1) << 30;               // 1 GiB per launch
            const size_t per_wg_vec = total / 16 / g / (size_t(t) * 8) * (size_t(t) * 8);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                stream_kernel<8><<<g, t>>>(buf, per_wg_vec, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double gb = double(per_wg_vec) * 16 * g / 1e9;
            printf("  {\"workgroups\": %d, \"threads\": %d, \"GBps\": %.0f, \"GBps_per_workgroup\": %.1f},\n", g, t, gb / (best * 1e-3), gb / (best * 1e-3) / g);
        }
    printf("  {}]}\n");
    return 0;
}
