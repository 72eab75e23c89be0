// kernels_comm.hip -- device side of the multi-GPU hand-over (include/qgym.h, "Multi-GPU hand-over").
//
// env.step() needs no exchange (envs are independent, rust/src/envs/clifford.rs:321-347 touches one env's state only).  What crosses
// GPUs is the learner's view of a shard after a step: the bit-packed observation (Env::observe, clifford.rs:361-368), the f32 reward
// (clifford.rs:355) and the is_final / success flags (clifford.rs:353,357-359), as ONE flat shard per rank:
//
//   [ obs: B * words_per_env * word_bytes | pad to 4 | reward: B * 4 | is_final: B | pad to 4 | success: B | pad to 16 ]
//
// Two transports move it: ncclAllGather (qgym_comm.cpp) and the direct write below -- every rank copies its shard straight into a
// window in each peer's HBM over xGMI (one link per peer: all seven links of a GPU carry one shard each at the same time, which
// is what the fully connected topology offers and a ring does not use) and raises a per-source flag; no collective library in the loop.
//
// Window (one per rank, uncached device memory shared by hipIpc): 4 KiB header {arrive[16], ack[16]} then 2 parities x world x stride.
//   arrive[s] = last epoch rank s has finished writing into THIS window        (written by s, polled by the owner)
//   ack[p]    = last epoch rank p has released in ITS window, stored HERE      (written by p, polled by the owner before it
//               overwrites the parity buffer of epoch - 2 in p's window)
// Every poll is bounded by a wall-clock deadline: a peer that never arrives raises an error word instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qgym_comm.hpp"

namespace qg {

__global__ __launch_bounds__(256) void shard_scalars_kernel(const float *__restrict__ reward, const uint8_t *__restrict__ done,
                                                           const uint8_t *__restrict__ success, uint8_t *__restrict__ shard, ShardLayout lay) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= lay.batch) return;
    if (e == 0) {  // the (at most 3 + 3 + 15) padding bytes are zero: a shard is a function of the env state, byte for byte
        for (uint64_t i = lay.obs_bytes; i < lay.reward_offset; ++i) shard[i] = 0;
        for (uint64_t i = lay.final_offset + lay.batch; i < lay.success_offset; ++i) shard[i] = 0;
        for (uint64_t i = lay.success_offset + lay.batch; i < lay.bytes; ++i) shard[i] = 0;
    }
    ((float *)(shard + lay.reward_offset))[e] = reward[e];
    shard[lay.final_offset + e] = done[e];
    shard[lay.success_offset + e] = success[e];
}

hipError_t shard_scalars(const float *reward, const uint8_t *done, const uint8_t *success, void *shard, const ShardLayout &lay, hipStream_t s) {
    const unsigned blocks = (unsigned)((lay.batch + 255) / 256);
    hipLaunchKernelGGL(shard_scalars_kernel, dim3(blocks), dim3(256), 0, s, reward, done, success, (uint8_t *)shard, lay);
    return hipGetLastError();
}

// ---- direct write -------------------------------------------------------------------------------------------------------------
__device__ inline uint32_t load_sys(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void store_sys(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// true when *p reached `want` before the deadline (wall_clock64 ticks at 100 MHz on gfx950)
__device__ inline bool poll_at_least(const uint32_t *p, uint32_t want, uint64_t timeout_ticks) {
    if ((int32_t)(load_sys(p) - want) >= 0) return true;
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < timeout_ticks) {
        __builtin_amdgcn_s_sleep(32);
        if ((int32_t)(load_sys(p) - want) >= 0) return true;
    }
    return false;
}

constexpr uint32_t PUSH_CHUNK16 = 1024;  // uint4 per block: 16 KiB, four per thread

// grid (chunks, world): block (c, y) copies chunk c of this rank's shard into the window of peer (rank + y) % world
__global__ __launch_bounds__(256) void push_shard_kernel(PushArgs a) {
    const uint32_t y = blockIdx.y;
    uint32_t peer = a.rank + y;
    if (peer >= a.world) peer -= a.world;
    __shared__ uint32_t go;
    if (threadIdx.x == 0) {
        // the parity buffer written now was last read in epoch - 2: the peer must have released it.  If it has not within the deadline,
        // this peer's copy AND its arrival flag are skipped -- the buffer may still be read there -- and the error word is raised: the
        // peer's wait then runs into its own deadline instead of reading a half-overwritten shard.  The error is sticky until the host
        // calls qg_comm_p2p_reset: while it is set no push writes to any peer and nothing waits (a dead peer costs one timeout, not one
        // per push and not one per wave of blocks).  Every block still reaches the ticket below.
        const bool down = __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & QG_COMM_ERR_ACK_TIMEOUT;
        const bool ok = !down && (a.epoch <= 2 || poll_at_least(a.local_ack + peer, a.epoch - 2, a.timeout_ticks));
        if (!ok && !down) atomicOr(a.error, QG_COMM_ERR_ACK_TIMEOUT);
        go = ok ? 1u : 0u;
    }
    __syncthreads();
    const bool copy = go != 0u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *__restrict__ src = (const u32x4 *)a.src;
    u32x4 *__restrict__ dst = (u32x4 *)a.dst[peer];
    const uint64_t base = (uint64_t)blockIdx.x * PUSH_CHUNK16;
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (copy && i < a.n16) v[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (copy && i < a.n16) __builtin_nontemporal_store(v[k], dst + i);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t *skipped = a.ticket + COMM_MAX_WORLD + y;  // set by a block that skipped its chunk
        if (!copy) atomicOr(skipped, 1u);
        __threadfence();
        const uint32_t taken = atomicAdd(a.ticket + y, 1u);
        if (taken == gridDim.x - 1) {  // the last block of this peer: every chunk is visible system-wide
            const bool whole = atomicExch(skipped, 0u) == 0u;
            a.ticket[y] = 0;
            __threadfence_system();
            if (whole) store_sys(a.arrive[peer] + a.rank, a.epoch);  // every chunk written: the shard has arrived
        }
    }
}

hipError_t push_shard(const PushArgs &a, hipStream_t s) {
    const unsigned chunks = (unsigned)((a.n16 + PUSH_CHUNK16 - 1) / PUSH_CHUNK16);
    hipLaunchKernelGGL(push_shard_kernel, dim3(chunks ? chunks : 1, a.world), dim3(256), 0, s, a);
    return hipGetLastError();
}

// one wave: lane s waits until rank s has written epoch `epoch` into this rank's window
__global__ __launch_bounds__(64) void wait_arrivals_kernel(const uint32_t *arrive, uint32_t world, uint32_t epoch, uint64_t timeout_ticks, uint32_t *error) {
    const uint32_t s = threadIdx.x;
    if (s < world && !poll_at_least(arrive + s, epoch, timeout_ticks)) atomicOr(error, QG_COMM_ERR_ARRIVE_TIMEOUT);
}

hipError_t wait_arrivals(const uint32_t *arrive, uint32_t world, uint32_t epoch, uint64_t timeout_ticks, uint32_t *error, hipStream_t s) {
    hipLaunchKernelGGL(wait_arrivals_kernel, dim3(1), dim3(64), 0, s, arrive, world, epoch, timeout_ticks, error);
    return hipGetLastError();
}

// one wave: lane p tells rank p that this rank has finished reading epoch `epoch` of its window
__global__ __launch_bounds__(64) void release_window_kernel(AckArgs a) {
    const uint32_t p = threadIdx.x;
    if (p < a.world) {
        __threadfence_system();
        store_sys(a.ack[p] + a.rank, a.epoch);
    }
}

hipError_t release_window(const AckArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(release_window_kernel, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace qg
