// qm_step1.hpp -- the body of the one-step kernel of the TILE layout (CliffordEnv N <= 16, LinearFunctionEnv 8 < N <= 32, no add_inverts),
// shared by qm_step1_kernel (kernels_qm.hip) and by the policy kernel that samples an action and steps its env in the same launch
// (mid_head_sample_kernel, kernels_policy.hip).  Reference: Clifford::step rust/src/envs/clifford.rs:321-347.
#pragma once

#include "device_common.hpp"

namespace qg {

#define QM_IDENTITY 0x8421u

template <bool HAS_Z>
__device__ inline void qm_group_get(const uint32_t (&u)[4], uint32_t q, uint32_t &x, uint32_t &z) {
    const uint32_t o = 0u - (q & 1u);
    if constexpr (HAS_Z) {  // {X[2g], Z[2g], X[2g+1], Z[2g+1]}
        x = (u[2] & o) | (u[0] & ~o);
        z = (u[3] & o) | (u[1] & ~o);
    } else {                // rows 4g .. 4g+3
        const uint32_t h = 0u - ((q >> 1) & 1u);
        const uint32_t lo = (u[1] & o) | (u[0] & ~o), hi = (u[3] & o) | (u[2] & ~o);
        x = (hi & h) | (lo & ~h);
        z = 0u;
    }
}
template <bool HAS_Z>
__device__ inline void qm_group_put(uint32_t (&u)[4], uint32_t q, uint32_t x, uint32_t z) {
    if constexpr (HAS_Z) {
        const uint32_t o = 0u - (q & 1u);
        u[0] = (u[0] & o) | (x & ~o); u[1] = (u[1] & o) | (z & ~o);
        u[2] = (x & o) | (u[2] & ~o); u[3] = (z & o) | (u[3] & ~o);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) u[k] = (q & 3u) == k ? x : u[k];
    }
}

// One env.step() of env `env` with action `act` (already loaded): gathers the gate's <= 2 row groups from the env's tile (G groups of
// 1 KiB), applies the gate's 4x4 GF(2) map, scatters, updates the incremental solved mask, depth, reward, done, success (and the
// solution log / layer metrics when FEAT).  Returns is_final.
template <bool HAS_Z, bool FEAT>
__device__ inline bool qm_step1_body(const StepArgs &a, uint32_t G, uint64_t env, int64_t act) {
    const uint32_t lane = (uint32_t)env & (QG_WAVE - 1);
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(G * 64);
    int32_t depth = a.depth[env];
    const uint32_t bad0 = a.bad[env];
    uint32_t bad = bad0, fault = 0;
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
    float penalty = 0.0f;
    if (in_range) {
        const GateEntry g = a.gates[act];
        penalty = g.penalty;
        const bool layered = FEAT && (a.flags & F_LAYERS);
        LayerTxn lt;
        if (layered) lt = layers_begin(layer_rec(a.layers, env, 2 * a.N + 2), a.N, a.descs[act]);  // its loads fly with the state's
        const uint32_t q0 = g.ops & 31u, q1 = (g.ops >> 5) & 31u, m = (g.ops >> 10) & 0xFFFFu;
        if (m != QM_IDENTITY) {  // "no gate" (e.g. a two-qubit gate on equal qubits) changes nothing
            constexpr uint32_t gsh = HAS_Z ? 1u : 2u;
            const uint32_t g0 = q0 >> gsh, g1 = q1 >> gsh;
            const uint4 va = tile[g0 * 64 + lane], vb = tile[g1 * 64 + lane];
            uint32_t ua[4] = {va.x, va.y, va.z, va.w}, ub[4] = {vb.x, vb.y, vb.z, vb.w};
            uint32_t x0, z0, x1, z1;
            qm_group_get<HAS_Z>(ua, q0, x0, z0);
            qm_group_get<HAS_Z>(ub, q1, x1, z1);
            auto mix = [&](uint32_t k) -> uint32_t {  // out_k = xor_i M[k][i] * in_i
                const uint32_t b = m >> (4 * k);
                uint32_t o = ((0u - (b & 1u)) & x0) ^ ((0u - ((b >> 2) & 1u)) & x1);
                if (HAS_Z) o ^= ((0u - ((b >> 1) & 1u)) & z0) ^ ((0u - ((b >> 3) & 1u)) & z1);
                return o;
            };
            const uint32_t nx0 = mix(0), nx1 = mix(2), nz0 = HAS_Z ? mix(1) : 0u, nz1 = HAS_Z ? mix(3) : 0u;
            // q1's rows first, then q0's (q0's value wins when q0 == q1, as in qm_apply)
            qm_group_put<HAS_Z>(ub, q1, nx1, nz1);
            const bool same = g0 == g1;
#pragma unroll
            for (int k = 0; k < 4; ++k) ua[k] = same ? ub[k] : ua[k];
            qm_group_put<HAS_Z>(ua, q0, nx0, nz0);
            if (!same) tile[g1 * 64 + lane] = make_uint4(ub[0], ub[1], ub[2], ub[3]);
            tile[g0 * 64 + lane] = make_uint4(ua[0], ua[1], ua[2], ua[3]);
            const uint32_t zb = 1u << a.N;
            const uint32_t b1 = (uint32_t)(nx1 != (1u << q1) || (HAS_Z && nz1 != (zb << q1)));
            const uint32_t b0 = (uint32_t)(nx0 != (1u << q0) || (HAS_Z && nz0 != (zb << q0)));
            bad = (bad & ~(1u << q1)) | (b1 << q1);
            bad = (bad & ~(1u << q0)) | (b0 << q0);
        }
        if (layered) penalty = layers_commit(lt, a.w);
    }
    if (FEAT && (a.flags & F_TRACK)) {  // clifford.rs:334-340
        if ((uint32_t)sol_n < a.sol_cap) sol_at(a, env, (uint32_t)sol_n++) = sol_word_framed(act, false);
        else fault |= 8u;
    }
    depth = depth > 0 ? depth - 1 : 0;  // clifford.rs:342
    const bool solved = bad == 0;       // clifford.rs:344
    const float achieved = solved ? 1.0f : 0.0f;
    const float reward = achieved - penalty;  // clifford.rs:345-346
    if (a.rewards_seq) a.rewards_seq[env] = reward;
    if (a.dones_seq) a.dones_seq[env] = (uint8_t)(depth == 0 || solved);
    if (bad != bad0) a.bad[env] = bad;
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);  // is_final (clifford.rs:353)
    a.success[env] = (uint8_t)solved;
    if (FEAT && (a.flags & F_TRACK)) a.sol_len[env * 2] = sol_n;
    if (FEAT && fault) atomicOr(&a.error[env], fault);
    return depth == 0 || solved;
}

}  // namespace qg
