// comm.hpp -- transport of the multi-GPU exchange step (friendship_render.h fr_shard).
//
// The render path has exactly one place where ranks exchange data: the partial-block reduce (and the optional gather
// of output rows to rank 0).  Both are pairwise: rank a sends a contiguous range of f32 to rank b and receives one.
// Two transports implement that:
//   * RCCL over xGMI (comm_rccl.cpp): the engine's own communicator, ncclSend/ncclRecv pairs on the call's stream,
//     device memory to device memory, no host involvement;
//   * a host callback (fr_comm): the engine stages the ranges through pinned host memory around it (engine.cpp).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <memory>

namespace fr {

struct Transport {
    virtual ~Transport() {}
    // Exchange with `peer`: n_send floats from d_send, n_recv floats into d_recv (device pointers; either count may
    // be 0), ordered on stream `st`.  The peer makes the mirror-image call.  Throws fr::Error(FR_ERR_COMM).
    virtual void sendrecv(uint32_t peer, const float *d_send, size_t n_send, float *d_recv, size_t n_recv, hipStream_t st) = 0;
    virtual const char *name() const = 0;
};

// ncclGetUniqueId / ncclCommInitRank.  Throw fr::Error(FR_ERR_COMM) when RCCL is unavailable or fails.
void rccl_unique_id(uint8_t id[128]);
std::unique_ptr<Transport> make_rccl_transport(const uint8_t id[128], uint32_t rank, uint32_t world);

}  // namespace fr
