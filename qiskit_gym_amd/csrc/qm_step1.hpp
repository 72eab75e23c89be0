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
// D16 > 0 (qg_vec_track_dense; the matrix has D = 16 * D16 rows, no padding slots): the rows the gate rewrote also go to the env's
// dense int8 observation -- <= 4 rows of D bytes instead of the D * D bytes a full qg_vec_observe_dense writes.
// `alone`: the neighbouring lane does not run this body with the neighbouring env (the reset's lanes in qm_reset_step_kernel): the dense rows are
// then written by this lane alone.
template <bool HAS_Z, bool FEAT, int D16 = 0>
__device__ inline bool qm_step1_body(const StepArgs &a, uint32_t G, uint64_t env, int64_t act, bool alone = false) {
    const uint32_t lane = (uint32_t)env & (QG_WAVE - 1);
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(G * 64);
    int32_t depth = a.depth[env];
    const uint32_t bad0 = a.bad[env];
    uint32_t bad = bad0, fault = 0;
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
    float penalty = 0.0f;
    uint32_t drow[4] = {0, 0, 0, 0}, dword[4] = {0, 0, 0, 0}, dchg = 0;  // D16: the rows the gate changed (bit k of dchg: entry k)
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
            if constexpr (D16 > 0) {  // the rows that changed (S rewrites one of a qubit's two rows, CX two of four, ...): stored below
                drow[0] = q0; dword[0] = nx0; dchg |= (uint32_t)(nx0 != x0);
                drow[1] = a.N + q0; dword[1] = nz0; dchg |= (uint32_t)(HAS_Z && nz0 != z0) << 1;
                drow[2] = q1; dword[2] = nx1; dchg |= (uint32_t)(q1 != q0 && nx1 != x1) << 2;
                drow[3] = a.N + q1; dword[3] = nz1; dchg |= (uint32_t)(HAS_Z && q1 != q0 && nz1 != z1) << 3;
            }
            const uint32_t zb = 1u << a.N;
            const uint32_t b1 = (uint32_t)(nx1 != (1u << q1) || (HAS_Z && nz1 != (zb << q1)));
            const uint32_t b0 = (uint32_t)(nx0 != (1u << q0) || (HAS_Z && nz0 != (zb << q0)));
            bad = (bad & ~(1u << q1)) | (b1 << q1);
            bad = (bad & ~(1u << q0)) | (b0 << q0);
        }
        if (layered) penalty = layers_commit(lt, a.w);
    }
    // qg_vec_track_dense: matrix row r, column c of env e at dense[(e * D + r) * D + c] (clifford.rs:361-368, adapters.py:50-54)
    if constexpr (D16 == 2) {
        // The two lanes of a pair (2k, 2k + 1: neighbouring envs) write one 32-byte row together, 16 bytes each, so that a store
        // instruction's lanes cover whole rows: per-lane rows (two 16-byte stores 16 bytes apart from ONE lane) measured 5.28 us per step at
        // 65 536 envs, this form 4.83 (3.11 without the dense observation).  Every lane that runs this body reaches this point; a pair lane
        // that does not (past the batch's odd end) reads as "no row" -- update_dpp keeps `old` = 0 for a disabled source lane -- and the
        // env it leaves without a partner writes its rows alone.
        const bool solo = alone || (env ^ 1ull) >= a.B;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (solo && ((dchg >> k) & 1u)) dense_row_store<2>(a.dense, env, drow[k], dword[k]);
            const uint32_t mine = (!solo && ((dchg >> k) & 1u)) ? (drow[k] | 0x80000000u) : 0u;
#pragma unroll
            for (int par = 0; par < 2; ++par) {  // quad_perm [0, 0, 2, 2] / [1, 1, 3, 3]: the even / odd lane's entry on both lanes of the pair
                const uint32_t r = par ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xF5, 0xF, 0xF, false)
                                       : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xA0, 0xF, 0xF, false);
                const uint32_t w = par ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dword[k], 0xF5, 0xF, 0xF, false)
                                       : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dword[k], 0xA0, 0xF, 0xF, false);
                const uint64_t e = (env & ~1ull) | (uint64_t)par;
                const uint32_t half = lane & 1u;
                if (r >> 31) *reinterpret_cast<uint4 *>(a.dense + (e * 32u + (r & 31u)) * 32u + 16u * half) = expand16_i8(w >> (16u * half));
            }
        }
    } else if constexpr (D16 == 1) {  // 16-byte rows: one store each
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((dchg >> k) & 1u) dense_row_store<1>(a.dense, env, drow[k], dword[k]);
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


// qg_vec_track_dense with add_inverts (N = 16): the envs of this wave whose matrix was inverted in this step, one after the other -- every lane
// writes its 16 bytes of the env's contiguous 1 KiB from the rows qm_inv2_body parked in `dense_lds` ([env of the wave][row], pitch 33).
// Call from all 64 lanes; `env0` = the wave's first env, `whole` false on lanes without an env.
__device__ inline void qm_inv2_dense_flush(int8_t *dense, const uint32_t *dense_lds, uint64_t env0, bool whole) {
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint64_t todo = __ballot(whole && !(lane & 1u));  // bit 2 * slot per inverted env of this wave
    while (todo) {  // wave-uniform
        const uint32_t sl = ((uint32_t)__ffsll((long long)todo) - 1u) >> 1;
        todo &= todo - 1ull;
        const uint32_t w = dense_lds[sl * 33u + (lane >> 1)];
        reinterpret_cast<uint4 *>(dense + (env0 + sl) * 1024u)[lane] = expand16_i8(w >> (16u * (lane & 1u)));
    }
}

// ---- the reference-default step of CliffordEnv N <= 16 (add_inverts), two lanes per env: see kernels_qm.hip for the discussion ----
#define QM_FLAG_INVERTED 1u
#define QM_FLAG_SYMPLECTIC 2u

// Balanced select tree: each level blends pairs with an all-ones/all-zeros lane mask, (hi & m) | (lo & ~m) = one v_bfi_b32; written
// as bit arithmetic on purpose: a `b ? t[2k+1] : t[2k]` select gets folded by the compiler into a runtime-indexed (scratch) array read.
template <int n>
__device__ inline uint32_t tree_select(const uint32_t (&t)[n], uint32_t q) {
    if constexpr (n == 1) {
        return t[0];
    } else {
        constexpr int m = (n + 1) / 2;
        uint32_t u[m];
        const uint32_t mb = 0u - (q & 1u);
#pragma unroll
        for (int k = 0; k < m; ++k) u[k] = (2 * k + 1 < n) ? ((t[2 * k + 1] & mb) | (t[2 * k] & ~mb)) : t[2 * k];
        return tree_select<m>(u, q >> 1);
    }
}

__device__ inline uint32_t qm_pair_swap(uint32_t v) {  // the partner lane's value (lanes 2e, 2e+1): DPP quad_perm [1, 0, 3, 2]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

// GS: the env's 16-byte groups (two qubits each) as a compile-time constant, or 0: `groups` at run time (the policy kernel).
// `env`, `h`: this lane's env and half (lanes 2e, 2e + 1 of a wave hold env e); `act`: the env's action, or null: load it from a.actions.
// Returns is_final (on both lanes).
// DENSE (qg_vec_track_dense, N = 16): `dense_lds` = 32 x 33 words of LDS owned by this wave; the env's dense int8 observation follows the step --
// the whole 1 KiB when the coin inverted the matrix (the wave's inverted envs one after the other, every lane 16 bytes of a contiguous 1 KiB),
// else the rows of the gate's qubits.
// The whole-env part needs all 64 lanes of the wave: the body only parks the rows (and reports `whole`), the kernel calls qm_inv2_dense_flush
// from EVERY lane afterwards, the ones past the batch's end included.
template <int GS, bool FEAT, bool DENSE = false>
__device__ inline bool qm_inv2_body(const StepArgs &a, uint32_t groups, uint64_t env, uint32_t h, const int64_t *act_in, uint32_t *dense_lds = nullptr,
                                    bool *dense_whole = nullptr) {
    const int G = GS ? GS : (int)groups;
    const uint32_t N = a.N;
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(G * 64) + (env & 63u);
    // every load that does not depend on another one is issued here, the rows first (the longest transfers), so that the whole
    // kernel pays two memory round trips: this batch, and the gate entry behind the action
    uint4 grp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        grp[k] = make_uint4(0u, 0u, 0u, 0u);
        if (k < G && (h == 0 || k + 4 < G)) grp[k] = tile[(uint32_t)(4 * h + k) * 64u];
    }
    int32_t depth = a.depth[env];
    uint32_t iflags = a.inverted[env];
    uint32_t coin = a.coins ? a.coins[env] : 0u;
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    int32_t sol_b = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2 + 1] : 0;
    const int64_t act = act_in ? *act_in : load_action(a.actions, env, a.flags & F_ACT64);
    const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
    GateEntry g = a.gates[in_range ? act : 0];  // unconditional (clamped) load: nothing else waits behind a branch
    if (!in_range) g = GateEntry{QM_IDENTITY << 10, 0.0f};
    if (!a.coins) coin = (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a)) >> 63);  // runs under the loads
    // this lane's 8 qubits: xs[j] = X row of qubit 8h + j, zs[j] = its Z row (stored words: bit c = logical column c)
    uint32_t xs[8], zs[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xs[2 * k] = grp[k].x; zs[2 * k] = grp[k].y; xs[2 * k + 1] = grp[k].z; zs[2 * k + 1] = grp[k].w;
    }
    uint32_t fault = 0;
    float penalty = g.penalty;
    if (FEAT && (a.flags & F_LAYERS) && in_range && h == 0) penalty = layers_penalty(layer_rec(a.layers, env, 2 * N + 2), N, a.descs[act], a.w);

    // ---- apply_gate_to_state (clifford.rs:331): the 4x4 GF(2) map on {X[q0], Z[q0], X[q1], Z[q1]} --------------------------
    uint32_t dirty = 0;  // this lane's groups that changed (bit k: group 4h + k)
    const uint32_t q0 = g.ops & 31u, q1 = (g.ops >> 5) & 31u, m = (g.ops >> 10) & 0xFFFFu;
    const bool own0 = (q0 >> 3) == h, own1 = (q1 >> 3) == h;
    {
        uint32_t x0 = own0 ? tree_select<8>(xs, q0 & 7u) : 0u, z0 = own0 ? tree_select<8>(zs, q0 & 7u) : 0u;
        uint32_t x1 = own1 ? tree_select<8>(xs, q1 & 7u) : 0u, z1 = own1 ? tree_select<8>(zs, q1 & 7u) : 0u;
        x0 |= qm_pair_swap(x0); z0 |= qm_pair_swap(z0);  // the lane that does not own the qubit contributes zero
        x1 |= qm_pair_swap(x1); z1 |= qm_pair_swap(z1);
        auto mix = [&](uint32_t k) -> uint32_t {  // out_k = xor_i M[k][i] * in_i
            const uint32_t b = m >> (4 * k);
            return ((0u - (b & 1u)) & x0) ^ ((0u - ((b >> 1) & 1u)) & z0) ^ ((0u - ((b >> 2) & 1u)) & x1) ^ ((0u - ((b >> 3) & 1u)) & z1);
        };
        const uint32_t nx0 = mix(0), nz0 = mix(1), nx1 = mix(2), nz1 = mix(3);
        if (m != QM_IDENTITY) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // q1's rows first, then q0's (q0's value wins when q0 == q1, as in qm_apply)
                const bool h1 = own1 && (q1 & 7u) == (uint32_t)j, h0 = own0 && (q0 & 7u) == (uint32_t)j;
                uint32_t vx = xs[j], vz = zs[j];
                vx = h1 ? nx1 : vx; vz = h1 ? nz1 : vz;
                vx = h0 ? nx0 : vx; vz = h0 ? nz0 : vz;
                xs[j] = vx; zs[j] = vz;
            }
            if (own0) dirty |= 1u << ((q0 & 7u) >> 1);
            if (own1) dirty |= 1u << ((q1 & 7u) >> 1);
        }
    }

    if (FEAT && (a.flags & F_TRACK) && h == 0) {  // clifford.rs:334-340: entries in push order, bit 31 = pushed to solution_inv
        if ((uint32_t)(sol_n + sol_b) < a.sol_cap) {
            const bool inv_frame = iflags & QM_FLAG_INVERTED;
            sol_at(a, env, (uint32_t)(sol_n + sol_b)) = sol_word_framed(act, inv_frame);
            if (inv_frame) ++sol_b;
            else ++sol_n;
        } else {
            fault |= 8u;
        }
    }
    depth = depth > 0 ? depth - 1 : 0;  // clifford.rs:342

    // ---- maybe_random_invert (clifford.rs:262-270) ----------------------------------------------------------------------------
    if (coin & 1u) {  // both lanes of a pair take the same branch
        if (iflags & QM_FLAG_SYMPLECTIC) {
            uint32_t w[16];  // local index i = 8 t + j  <->  v = 16 t + 8 h + j
            const uint32_t xm = N >= 16 ? 0xFFFFu : ((1u << N) - 1u);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                w[j] = xs[j];
                w[8 + j] = zs[j];
            }
            if (N < 16) {  // logical Z columns N .. 2N-1 move to bit positions 16 .. 16+N-1
#pragma unroll
                for (int i = 0; i < 16; ++i) w[i] = (w[i] & xm) | (((w[i] >> N) & xm) << 16);
            }
            // stages 4, 2, 1: word pairs (i, i + d) of this lane
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const int d = 4 >> st;
                const uint32_t mlo = st == 0 ? 0x0F0F0F0Fu : st == 1 ? 0x33333333u : 0x55555555u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if ((i & d) == 0) {
                        const uint32_t lo = w[i], hi = w[i + d];
                        w[i] = (lo & mlo) | ((hi & mlo) << d);
                        w[i + d] = ((lo >> d) & mlo) | (hi & ~mlo);
                    }
                }
            }
            // stage 8: the partner lane holds the other word of every pair; byte 1 / 3 of the low word <-> byte 0 / 2 of the high word
            const uint32_t sel8 = h ? 0x03070105u : 0x06020400u;
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = __builtin_amdgcn_perm(qm_pair_swap(w[i]), w[i], sel8);
            // stage 16 + Omega: inv[v] = rot16(T[v ^ 16])
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t lo = w[j], hi = w[8 + j];
                w[j] = __builtin_amdgcn_perm(hi, lo, 0x03020706u);      // {hi.hi16, lo.hi16}
                w[8 + j] = __builtin_amdgcn_perm(hi, lo, 0x01000504u);  // {hi.lo16, lo.lo16}
            }
            if (N < 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) w[i] = (w[i] & xm) | (((w[i] >> 16) & xm) << N);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xs[j] = w[j];
                zs[j] = w[8 + j];
            }
            iflags ^= QM_FLAG_INVERTED;
            dirty = 0xFu;
        } else {
            fault |= QG_FAULT_BAD_STATE;  // unreachable: the host launches the Gauss-Jordan variant whenever such an env may exist
        }
    }

    // ---- solved (clifford.rs:344), reward (:345-346) ----------------------------------------------------------------------------
    uint32_t diff = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t q = 8u * h + (uint32_t)j;
        const bool real = q < N;
        diff |= xs[j] ^ (real ? 1u << q : 0u);
        diff |= zs[j] ^ (real ? (1u << N) << q : 0u);
    }
    diff |= qm_pair_swap(diff);
    const bool solved = diff == 0;
    const float achieved = solved ? 1.0f : 0.0f;
    const float reward = achieved - penalty;

#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (((dirty >> k) & 1u) && k < G && (h == 0 || k + 4 < G))
            tile[(uint32_t)(4 * h + k) * 64u] = make_uint4(xs[2 * k], zs[2 * k], xs[2 * k + 1], zs[2 * k + 1]);
    if constexpr (DENSE) {  // N = 16: matrix row r of env e at dense[(e * 32 + r) * 32] (clifford.rs:361-368, adapters.py:50-54)
        const uint32_t lane = threadIdx.x & (QG_WAVE - 1), slot = lane >> 1;
        const bool whole = dirty == 0xFu;  // the inversion rewrote every row (both lanes of the pair agree)
        if (whole) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dense_lds[slot * 33u + 8u * h + (uint32_t)j] = xs[j];
                dense_lds[slot * 33u + 16u + 8u * h + (uint32_t)j] = zs[j];
            }
        }
        *dense_whole = whole;
        if (!whole && m != QM_IDENTITY) {  // the gate's rows only (this lane's share of them)
            if (own0) {
                dense_row_store<2>(a.dense, env, q0, tree_select<8>(xs, q0 & 7u));
                dense_row_store<2>(a.dense, env, 16u + q0, tree_select<8>(zs, q0 & 7u));
            }
            if (own1 && q1 != q0) {
                dense_row_store<2>(a.dense, env, q1, tree_select<8>(xs, q1 & 7u));
                dense_row_store<2>(a.dense, env, 16u + q1, tree_select<8>(zs, q1 & 7u));
            }
        }
    }
    if (h == 0) {
        if (a.rewards_seq) a.rewards_seq[env] = reward;
        if (a.dones_seq) a.dones_seq[env] = (uint8_t)(depth == 0 || solved);
        a.depth[env] = depth;
        a.reward[env] = reward;
        a.done[env] = (uint8_t)(depth == 0 || solved);  // is_final (clifford.rs:353)
        a.success[env] = (uint8_t)solved;
        if (FEAT && (a.flags & F_TRACK)) {
            a.sol_len[env * 2] = sol_n;
            a.sol_len[env * 2 + 1] = sol_b;
        }
        a.inverted[env] = (uint8_t)iflags;
        if (fault) atomicOr(&a.error[env], fault);
    }
    return depth == 0 || solved;
}

}  // namespace qg
