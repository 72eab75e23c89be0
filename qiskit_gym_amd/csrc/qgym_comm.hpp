// qgym_comm.hpp -- private definitions of the multi-GPU hand-over (qgym_comm.cpp, kernels_comm.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qgym.h"

namespace qg {

constexpr uint32_t COMM_MAX_WORLD = 16;
constexpr uint64_t COMM_HEADER_BYTES = 4096;  // window header: arrive[16] at 0, ack[16] at 256
constexpr uint64_t COMM_ACK_OFFSET = 256;

enum : uint32_t { QG_COMM_ERR_ACK_TIMEOUT = 1u, QG_COMM_ERR_ARRIVE_TIMEOUT = 2u };

// byte offsets inside one rank's flat shard (qg_shard_layout with the batch)
struct ShardLayout {
    uint64_t batch;
    uint64_t bytes;
    uint64_t obs_bytes;
    uint64_t reward_offset;
    uint64_t final_offset;
    uint64_t success_offset;
};

struct PushArgs {
    const uint4 *src;             // this rank's packed shard (local)
    uint64_t n16;                 // its length in 16-byte units
    uint4 *dst[COMM_MAX_WORLD];   // slot `rank` of this epoch's parity buffer in every rank's window
    uint32_t *arrive[COMM_MAX_WORLD];  // arrive[] array of every rank's window
    const uint32_t *local_ack;    // ack[] array of this rank's own window
    uint32_t *ticket;             // [world] block counters (local, zero between launches)
    uint32_t *error;              // local error word
    uint64_t timeout_ticks;
    uint32_t rank, world, epoch;
};

struct AckArgs {
    uint32_t *ack[COMM_MAX_WORLD];  // ack[] array of every rank's window
    uint32_t rank, world, epoch;
};

hipError_t shard_scalars(const float *reward, const uint8_t *done, const uint8_t *success, void *shard, const ShardLayout &lay, hipStream_t s);
hipError_t push_shard(const PushArgs &a, hipStream_t s);
hipError_t wait_arrivals(const uint32_t *arrive, uint32_t world, uint32_t epoch, uint64_t timeout_ticks, uint32_t *error, hipStream_t s);
hipError_t release_window(const AckArgs &a, hipStream_t s);

}  // namespace qg
