// pbl_comm.hip -- one-shot all-reduce of the K-split partial outputs over peer-mapped buffers.
//
// The reference has no multi-GPU code (evaluate.py:56-62 "TODO: fix multi-gpu").  A K-split PB linear ends in a sum of
// [M, N] fp32 partials over the ranks; at decode time that is 16 KB - 1 MB, far below the size where a ring collective
// is bandwidth bound: RCCL's all-reduce costs tens of microseconds of protocol latency next to a ~1 us GEMV.  xGMI is
// point to point (every GPU has a direct link to each of its 7 peers), so the latency-optimal exchange is ONE hop:
//
//   every rank r owns a buffer  { flags[2][16][B], control words, data[2][P][cap] }  that all ranks have mapped (hipIpc, one
//   process per GPU).  all-reduce number `seq` (set = seq & 1), block b of the launch, slice [lo, hi) of the vector:
//     1. push:   for every peer p:  p.data[set][r][lo:hi] <- x[lo:hi]            (7 direct xGMI writes + 1 local)
//     2. signal: system-scope release, then  p.flags[set][r][b] <- seq            (one 4-byte store per peer)
//     3. wait:   until  own.flags[set][p][b] == seq  for every p                  (bounded spin, system-scope acquire)
//     4. reduce: x[i] <- sum_p own.data[set][p][i]   in rank order                (identical bits on every rank)
//   A slot of set s is rewritten at seq + 2; a peer can only be there after it has seen THIS rank's flag of seq + 1,
//   which this rank raises after finishing step 4 of seq -- two sets are enough, no extra barrier.
//   Blocks are independent (block b waits only for block b of the peers), so nothing requires co-residency.
//   `seq` is either a launch argument (pbl_p2p_allreduce_f32) or -- pbl_p2p_allreduce_f32_dev -- kept in the buffer itself:
//   every block reads "last finished call + 1", the last block to finish publishes it.  The launch then has no per-call
//   argument at all and can be captured in a hipGraph next to the K-split GEMV and replayed.  A wait that times out
//   (3 s: a peer died or never launched) sets the status word AND writes NaN over the slice: a wrong sum is never silent.
//
// The buffers are allocated here (the one place libpbl allocates: a peer-mapped buffer must be its own hipMalloc
// allocation so that its IPC handle maps exactly it), uncached, so peer writes and local polls never sit in an L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbl.h"
#include "pbl_p2p_layout.h"

namespace {

using namespace pblp2p;   // buffer layout: csrc/pbl_p2p_layout.h (shared with the K-split GEMV's fused push, csrc/pbl_kernels.hip)

struct P2PArgs {
    uint8_t* peer[MAXW];     // this process's mapping of every rank's buffer (peer[rank] = own)
    float* x;                // in / out
    _Float16* y16;           // optional second result, rounded to fp16 (the K-split layer's output dtype): saves a cast launch
    size_t n, cap;
    uint32_t seq;            // 0: take the call number from the buffer's own counter (hipGraph-replayable)
    int rank, world, nblk;
};

__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PArgs a) {
    __shared__ uint32_t s_seq, s_timeout;
    __shared__ uint64_t s_limit;
    const int b = blockIdx.x;
    uint32_t* ctl = ctl_ptr(a.peer[a.rank]);
    if (threadIdx.x == 0) {
        // the call number: an argument, or one more than the last call this buffer finished.  Every block of the launch reads
        // the same value: ctl[1] is only advanced by the LAST block to finish, and launches on a stream do not overlap.
        uint32_t q = a.seq ? a.seq : __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s_seq = q ? q : 1u;                                                      // 0 is the flags' idle value
        s_timeout = 0;
        // once a wait has timed out on this buffer (status word set: a peer is gone, pbl_p2p_check raises) later calls give up
        // after 1 ms instead of 3 s each -- a dead peer costs one long wait, not one per layer and token
        s_limit = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 100000ull : 300000000ull;
    }
    __syncthreads();
    const uint32_t seq = s_seq;
    const int set = int(seq & 1u);
    const size_t per = ((a.n + a.nblk - 1) / a.nblk + 3) & ~size_t(3);           // slice length, multiple of 4 floats
    const size_t lo = size_t(b) * per, hi = lo + per < a.n ? lo + per : a.n;
    const bool vec = ((reinterpret_cast<uintptr_t>(a.x) | (a.cap * 4)) & 15) == 0;
    // 1. push my slice to every rank's slot [set][rank]
    for (int p = 0; p < a.world; ++p) {
        float* dst = slot_ptr(a.peer[p], set, a.rank, a.world, a.cap);
        if (vec) {
            for (size_t i = lo + size_t(threadIdx.x) * 4; i < hi; i += 1024) {
                if (i + 4 <= hi) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(a.x + i);
                else for (size_t j = i; j < hi; ++j) dst[j] = a.x[j];
            }
        } else {
            for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = a.x[i];
        }
    }
    // 2. make the slice visible system wide, then raise my flag at every rank
    __threadfence_system();
    __syncthreads();
    if (int(threadIdx.x) < a.world)
        __hip_atomic_store(flag_ptr(a.peer[threadIdx.x], set, a.rank, b), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 3. wait for every rank's flag (bounded: a lost peer must not hang the GPU; the status word records it)
    if (int(threadIdx.x) < a.world) {
        const uint32_t* f = flag_ptr(a.peer[a.rank], set, int(threadIdx.x), b);
        const uint64_t t0 = wall_clock64();                                          // 100 MHz
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > s_limit) {                                          // 3 s (1 ms once the buffer has seen a time-out)
                atomicExch(ctl, 1u);
                s_timeout = 1;
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    // 4. sum the P slots in rank order; a timed-out wait POISONS the slice instead of summing stale slots
    const uint8_t* own = a.peer[a.rank];
    const bool bad = s_timeout != 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        float s = 0.f;
        for (int p = 0; p < a.world; ++p)
            s += __builtin_nontemporal_load(slot_ptr(const_cast<uint8_t*>(own), set, p, a.world, a.cap) + i);
        if (bad) s = __builtin_nanf("");
        a.x[i] = s;
        if (a.y16) a.y16[i] = _Float16(s);
    }
    // 5. device-counted calls: the last block to finish publishes the call number for the next launch
    if (a.seq == 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const uint32_t done = atomicAdd(ctl + 2, 1u);
            if (done == uint32_t(a.nblk) - 1u) {
                __hip_atomic_store(ctl + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctl + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The second half of a FUSED K-split layer (round 4): the GEMV's row owners have written their fp32 partials straight into every
// rank's slot [set][rank] and counted their records at count[set][rank] (pbl_linear_f16_push, csrc/pbl_kernels.hip); this
// kernel only waits until every rank has pushed `expect` records, sums the slots in rank order and publishes the call number.
// Compared with GEMV -> p2p_allreduce_kernel the partial never makes the round trip through local HBM and the push costs no
// launch of its own.  The last block to finish zeroes the counters of its set: a peer can only push into this set again after it
// has seen THIS rank's push of the next call, which comes after this kernel (same argument as for the slots).
__global__ __launch_bounds__(256) void p2p_reduce_kernel(P2PArgs a, uint32_t expect) {
    __shared__ uint32_t s_seq, s_timeout;
    __shared__ uint64_t s_limit;
    const int b = blockIdx.x;
    uint8_t* own = a.peer[a.rank];
    uint32_t* ctl = ctl_ptr(own);
    if (threadIdx.x == 0) {
        const uint32_t q = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s_seq = q ? q : 1u;
        s_timeout = 0;
        s_limit = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 100000ull : 300000000ull;     // (see p2p_allreduce_kernel)
    }
    __syncthreads();
    const uint32_t seq = s_seq;
    const int set = int(seq & 1u);
    const size_t per = ((a.n + a.nblk - 1) / a.nblk + 3) & ~size_t(3);
    const size_t lo = size_t(b) * per, hi = lo + per < a.n ? lo + per : a.n;
    if (int(threadIdx.x) < a.world) {
        const uint32_t* c = count_ptr(own, set, int(threadIdx.x));
        const uint64_t t0 = wall_clock64();                                          // 100 MHz
        while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < expect) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > s_limit) {                                          // 3 s (1 ms once the buffer has seen a time-out)
                atomicExch(ctl, 1u);
                s_timeout = 1;
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    const bool bad = s_timeout != 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        float s = 0.f;
        for (int p = 0; p < a.world; ++p) s += __builtin_nontemporal_load(slot_ptr(own, set, p, a.world, a.cap) + i);
        if (bad) s = __builtin_nanf("");
        if (a.x) a.x[i] = s;
        if (a.y16) a.y16[i] = _Float16(s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t done = atomicAdd(ctl + 2, 1u);
        if (done == uint32_t(a.nblk) - 1u) {
            for (int p = 0; p < a.world; ++p) __hip_atomic_store(count_ptr(own, set, p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(ctl + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int launch_allreduce(void* const* peer_bufs, int rank, int world, float* x, void* y16, size_t n, size_t max_elems, uint32_t seq,
                     void* stream) {
    if (!peer_bufs || !x || world < 1 || world > MAXW || rank < 0 || rank >= world || !n) return PBL_ERR_INVALID_ARG;
    const size_t cap = (max_elems + 3) & ~size_t(3);
    if (n > cap) return PBL_ERR_CAPACITY;
    P2PArgs a;
    for (int p = 0; p < MAXW; ++p) a.peer[p] = p < world ? static_cast<uint8_t*>(peer_bufs[p]) : nullptr;
    for (int p = 0; p < world; ++p) if (!a.peer[p]) return PBL_ERR_INVALID_ARG;
    a.x = x; a.y16 = static_cast<_Float16*>(y16); a.n = n; a.cap = cap; a.seq = seq; a.rank = rank; a.world = world;
    a.nblk = int((n + 4095) / 4096);
    if (a.nblk > MAXB) a.nblk = MAXB;
    if (a.nblk < 1) a.nblk = 1;
    void* argv[] = {&a};
    return hipLaunchKernel(reinterpret_cast<const void*>(p2p_allreduce_kernel), dim3(a.nblk), dim3(256), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

}  // namespace

extern "C" {

size_t pbl_p2p_buffer_bytes_world(size_t max_elems, int world) {
    if (world < 1 || world > MAXW) return 0;
    const size_t cap = (max_elems + 3) & ~size_t(3);
    return HDR_BYTES + flags_bytes() + size_t(2) * size_t(world) * cap * sizeof(float);
}
size_t pbl_p2p_buffer_bytes(size_t max_elems) { return pbl_p2p_buffer_bytes_world(max_elems, MAXW); }

int pbl_comm_alloc(size_t bytes, void** out) {
    if (!out || !bytes) return PBL_ERR_INVALID_ARG;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&p, bytes) != hipSuccess) return PBL_ERR_CAPACITY;
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return PBL_ERR_LAUNCH; }
    *out = p;
    return PBL_OK;
}

int pbl_comm_free(void* p) { return (!p || hipFree(p) == hipSuccess) ? PBL_OK : PBL_ERR_INVALID_ARG; }

int pbl_ipc_export(void* dev_ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == PBL_IPC_HANDLE_BYTES, "handle size");
    if (!dev_ptr || !handle64) return PBL_ERR_INVALID_ARG;
    return hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle64), dev_ptr) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

int pbl_ipc_open(const void* handle64, void** out) {
    if (!handle64 || !out) return PBL_ERR_INVALID_ARG;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle64, sizeof(h));
    return hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

int pbl_ipc_close(void* p) { return (p && hipIpcCloseMemHandle(p) == hipSuccess) ? PBL_OK : PBL_ERR_INVALID_ARG; }

int pbl_p2p_allreduce_f32(void* const* peer_bufs, int rank, int world, float* x, size_t n, size_t max_elems, uint32_t seq,
                          void* stream) {
    if (seq == 0) return PBL_ERR_INVALID_ARG;
    return launch_allreduce(peer_bufs, rank, world, x, nullptr, n, max_elems, seq, stream);
}

int pbl_p2p_allreduce_f32_dev(void* const* peer_bufs, int rank, int world, float* x, void* y_f16, size_t n, size_t max_elems,
                              void* stream) {
    return launch_allreduce(peer_bufs, rank, world, x, y_f16, n, max_elems, 0u, stream);
}

int pbl_p2p_reduce_f32_dev(void* const* peer_bufs, int rank, int world, float* y_f32, void* y_f16, size_t n, size_t max_elems,
                           uint32_t expect_records, void* stream) {
    if (!peer_bufs || (!y_f32 && !y_f16) || world < 1 || world > MAXW || rank < 0 || rank >= world || !n || !expect_records) return PBL_ERR_INVALID_ARG;
    const size_t cap = (max_elems + 3) & ~size_t(3);
    if (n > cap) return PBL_ERR_CAPACITY;
    P2PArgs a;
    for (int p = 0; p < MAXW; ++p) a.peer[p] = p < world ? static_cast<uint8_t*>(peer_bufs[p]) : nullptr;
    for (int p = 0; p < world; ++p) if (!a.peer[p]) return PBL_ERR_INVALID_ARG;
    a.x = y_f32; a.y16 = static_cast<_Float16*>(y_f16); a.n = n; a.cap = cap; a.seq = 0; a.rank = rank; a.world = world;
    a.nblk = int((n + 4095) / 4096);
    if (a.nblk > MAXB) a.nblk = MAXB;
    if (a.nblk < 1) a.nblk = 1;
    void* argv[] = {&a, &expect_records};
    return hipLaunchKernel(reinterpret_cast<const void*>(p2p_reduce_kernel), dim3(a.nblk), dim3(256), argv, 0,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? PBL_OK : PBL_ERR_LAUNCH;
}

int pbl_p2p_check(const void* own_buf) {
    // debugging aid, SYNCHRONOUS: reads the status word behind the flags (1 = a wait timed out since the buffer was created)
    if (!own_buf) return PBL_ERR_INVALID_ARG;
    // The word is cleared once it has been reported (it used to stay set for the buffer's life, so ONE transient stall --
    // a first-use kernel load, a host hiccup on one rank -- left every later wait at the 1 ms limit and ordinary launch skew then
    // poisoned results until the process ended): the caller raises for the call that timed out, later calls wait 3 s again.
    uint32_t w = 0;
    uint8_t* sp = const_cast<uint8_t*>(static_cast<const uint8_t*>(own_buf)) + flags_bytes();
    if (hipMemcpy(&w, sp, 4, hipMemcpyDeviceToHost) != hipSuccess) return PBL_ERR_LAUNCH;
    if (w) { const uint32_t z = 0; if (hipMemcpy(sp, &z, 4, hipMemcpyHostToDevice) != hipSuccess) return PBL_ERR_LAUNCH; }
    return int(w);
}

}  // extern "C"
