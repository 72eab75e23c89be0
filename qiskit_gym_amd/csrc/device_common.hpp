// device_common.hpp -- device helpers shared by the step kernels (gfx950, wave64).
#pragma once

#include "qgym_internal.hpp"

namespace qg {

#define QG_WAVE 64

__device__ inline int64_t load_action(const void *actions, uint64_t idx, bool act64) {
    return act64 ? reinterpret_cast<const int64_t *>(actions)[idx]
                 : (int64_t) reinterpret_cast<const int32_t *>(actions)[idx];
}

// Length of the compacted list of finished envs (compact_done), read by every thread of the LAST kernel
// that consumes it; the last block to have read it zeroes the counter (and the ticket) for the next
// qg_vec_reset_done.  counter[0] = length, counter[1] = blocks that have read it.  Call from all threads.
__device__ inline uint32_t list_count_take(uint32_t *counter) {
    const uint32_t count = counter[0];
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&counter[1], 1u) == gridDim.x - 1u) {
        counter[0] = 0;
        counter[1] = 0;
    }
    return count;
}

// Solution-log word of an action (clifford.rs:334-340 pushes the action verbatim, valid or not):
// the log is 32-bit, so anything a `usize` action could hold beyond 2^32 - 2 -- and a negative
// int64, which `as usize` turns into 2^64 - 1 -- saturates to 0xFFFFFFFF (read back as UINT64_MAX).
__device__ inline uint32_t sol_word(int64_t act) { return (act < 0 || act > 0xFFFFFFFEll) ? 0xFFFFFFFFu : (uint32_t)act; }

// Dense {0,1} elements from packed bits: one 16-byte chunk = 16 / ES elements of ES bytes each.
// `one`: the bit pattern of 1 in the output dtype (int8 1, bf16 0x3F80, f16 0x3C00, f32 0x3F800000)
template <int ES>
__device__ inline uint4 expand_chunk(uint32_t bits, uint32_t one) {
    uint32_t w[4];
    if constexpr (ES == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t nb = (bits >> (4 * k)) & 0xFu;
            w[k] = ((nb & 1u) | ((nb & 2u) << 7) | ((nb & 4u) << 14) | ((nb & 8u) << 21)) * one;
        }
    } else if constexpr (ES == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = (bits >> (2 * k)) & 1u, hi = (bits >> (2 * k + 1)) & 1u;
            w[k] = (lo * one) | ((hi * one) << 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = ((bits >> k) & 1u) * one;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// MetricsTracker with running maxima instead of HashSets (metrics.rs:83-123).
// `lay` = this env's record: last_gates[N], last_cxs[N], n_layers, n_layers_cnots (int32 each;
// last_* start at -1).  |layers| == max(last_gates)+1 and |cnot_layers| == max(last_cxs)+1
// because every inserted layer index is one more than an index already present (or 0).
struct LayerDelta {
    int dc, dlc, dl, dg;
};
__device__ inline void layers_single(int32_t *lay, uint32_t N, uint32_t t, LayerDelta &d) {
    if (t >= N) return;  // metrics.rs:84-86
    d.dg += 1;
    int32_t gl = lay[t] + 1;
    lay[t] = gl;
    int32_t nl = lay[2 * N];
    if (gl + 1 > nl) {
        d.dl += gl + 1 - nl;
        lay[2 * N] = gl + 1;
    }
}
__device__ inline void layers_cx(int32_t *lay, uint32_t N, uint32_t c, uint32_t t, LayerDelta &d) {
    if (c == t || c >= N || t >= N) return;  // metrics.rs:98-103
    d.dc += 1;
    d.dg += 1;
    int32_t a = lay[c], b = lay[t];
    int32_t gl = (a > b ? a : b) + 1;
    lay[c] = gl;
    lay[t] = gl;
    int32_t nl = lay[2 * N];
    if (gl + 1 > nl) {
        d.dl += gl + 1 - nl;
        lay[2 * N] = gl + 1;
    }
    a = lay[N + c];
    b = lay[N + t];
    int32_t cl = (a > b ? a : b) + 1;
    lay[N + c] = cl;
    lay[N + t] = cl;
    int32_t nlc = lay[2 * N + 1];
    if (cl + 1 > nlc) {
        d.dlc += cl + 1 - nlc;
        lay[2 * N + 1] = cl + 1;
    }
}
// metrics.rs:64-81 + 135-146: apply the gate to the tracker, return the f32 penalty.
// (This translation unit is compiled with -ffp-contract=off: no fused multiply-add.)
__device__ inline float layers_penalty(int32_t *lay, uint32_t N, uint32_t desc, const float w[4]) {
    uint32_t kind = desc & 0xFFu, q0 = (desc >> 8) & 0xFFu, q1 = (desc >> 16) & 0xFFu;
    LayerDelta d = {0, 0, 0, 0};
    switch (kind) {
    case QG_CX: layers_cx(lay, N, q0, q1, d); break;
    case QG_SWAP:
        layers_cx(lay, N, q0, q1, d);
        layers_cx(lay, N, q1, q0, d);
        layers_cx(lay, N, q0, q1, d);
        break;
    case QG_CZ:
        layers_single(lay, N, q1, d);
        layers_cx(lay, N, q0, q1, d);
        layers_single(lay, N, q1, d);
        break;
    default: layers_single(lay, N, q0, d); break;
    }
    float t0 = w[0] * (float)d.dc;
    float t1 = w[1] * (float)d.dlc;
    float t2 = w[2] * (float)d.dl;
    float t3 = w[3] * (float)d.dg;
    float s = t0 + t1;
    s = s + t2;
    s = s + t3;
    return s;
}

}  // namespace qg
