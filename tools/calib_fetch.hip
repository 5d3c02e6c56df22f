// FETCH_SIZE calibration: stream-read a known number of bytes with 16 B/lane nontemporal
// loads (the access width of the PB GEMV weight stream), so that the gfx950 FETCH_SIZE
// under-count (MI355X_MICROARCH.md, HBM section) can be measured in the same rocprofv3 run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void calib_stream_read(const u32x4* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        u32x4 v = __builtin_nontemporal_load(p + i);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// the same bytes read 2 B per lane (the access width of the GEMV's col0 / tail-count loads)
__global__ void calib_narrow_read(const uint16_t* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
// ... and 4 B per lane
__global__ void calib_dword_read(const uint32_t* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = size_t(2) << 30;  // 2 GiB > Infinity Cache
    void* buf; uint32_t* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 0x5A, bytes);
    for (int i = 0; i < 5; ++i) calib_stream_read<<<256 * 8, 256>>>((const u32x4*)buf, bytes / 16, out);
    hipDeviceSynchronize();
    const size_t nb = size_t(512) << 20;   // 512 MiB for the narrow variants
    for (int i = 0; i < 3; ++i) calib_narrow_read<<<256 * 8, 256>>>((const uint16_t*)buf, nb / 2, out);
    for (int i = 0; i < 3; ++i) calib_dword_read<<<256 * 8, 256>>>((const uint32_t*)buf, nb / 4, out);
    hipDeviceSynchronize();
    printf("calib_stream_read: %zu bytes per dispatch\n", bytes);
    return 0;
}
