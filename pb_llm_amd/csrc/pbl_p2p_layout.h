// pbl_p2p_layout.h -- layout of a peer-mapped communication buffer (csrc/pbl_comm.hip owns it; csrc/pbl_kernels.hip writes into
// it from the K-split GEMV's epilogue).  See pbl_comm.hip for the protocol.
//   [ flags[2][MAXW][MAXB] u32 (8 KiB) | control words (4 KiB) | data[2][world][cap] floats ]
//   ctl[0] status (1: a wait timed out), ctl[1] call number of the last finished device-counted call, ctl[2] blocks finished in
//   the running call, ctl[8 + 16 set + src]: records rank `src` has pushed into this buffer's slot [set][src] (fused push).
#ifndef PBL_P2P_LAYOUT_H_
#define PBL_P2P_LAYOUT_H_
#include <stdint.h>

#include "../../include/pbl.h"

namespace pblp2p {
constexpr int MAXB = PBL_P2P_MAX_BLOCKS;
constexpr int MAXW = PBL_P2P_MAX_WORLD;
constexpr size_t HDR_BYTES = 4096;
__host__ __device__ inline size_t flags_bytes() { return size_t(2) * MAXW * MAXB * 4; }
__device__ __forceinline__ uint32_t* flag_ptr(uint8_t* buf, int set, int src, int b) {
    return reinterpret_cast<uint32_t*>(buf) + (size_t(set) * MAXW + src) * MAXB + b;
}
__device__ __forceinline__ uint32_t* ctl_ptr(uint8_t* buf) { return reinterpret_cast<uint32_t*>(buf + flags_bytes()); }
__device__ __forceinline__ uint32_t* count_ptr(uint8_t* buf, int set, int src) { return ctl_ptr(buf) + 8 + set * MAXW + src; }
__device__ __forceinline__ float* slot_ptr(uint8_t* buf, int set, int src, int world, size_t cap) {
    return reinterpret_cast<float*>(buf + HDR_BYTES + flags_bytes()) + (size_t(set) * world + src) * cap;
}
}  // namespace pblp2p
#endif
