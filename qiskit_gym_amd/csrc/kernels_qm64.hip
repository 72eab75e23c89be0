// kernels_qm64.hip -- TILE64 layout: the thread-per-env "qubit machine" of kernels_qm.hip for
// matrices whose rows need 64-bit words: CliffordEnv 16 < N <= 32 (with or without add_inverts)
// and LinearFunctionEnv 32 < N <= 64 without inverts.
//
// Reference semantics: Clifford::step rust/src/envs/clifford.rs:321-347 (gates :89-133, solved
// :136-145, inverse :147-170, maybe_random_invert :262-270); LinearFunction::step
// rust/src/envs/linear_function.rs:302-328 (cx/swap :62-83).
//
// Same design as kernels_qm.hip (one lane owns one env, rows in VGPRs, an action is (q0, q1, 4x4
// GF(2) matrix) on {X[q0], Z[q0], X[q1], Z[q1]}, nothing crosses lanes); rows are uint64 and a
// 16-byte group holds two of them.  Layout: tiles of 64 envs, NS/2 groups of 1 KiB per tile, group g
// = for lane l the uint4 {slot 2g, slot 2g+1}.  CliffordEnv: slot 2j = X-type row j, slot 2j+1 =
// Z-type row N+j, i.e. one group per qubit -- a one-qubit gate dirties one group, a two-qubit gate
// two.  LinearFunctionEnv: slot j = row j.  The lane-group ROWS kernels these shapes used before
// are instruction-issue-bound (14-15 us per step, 140-220 us with inversion at B = 65 536).
#include "device_common.hpp"
#include "qgym_plan.hpp"

namespace qg {

// ops word: [0:6) q0, [6:12) q1, [12:28) M (bit 4k+i: output k takes input i, order X0,Z0,X1,Z1)
#define Q64_IDENTITY 0x8421u
#define Q64_FLAG_INVERTED 1u
#define Q64_FLAG_SYMPLECTIC 2u

template <int NS>
struct Q64Rows {
    static constexpr int G = NS / 2;
    uint64_t r[NS];
};

template <int n>
__device__ inline uint64_t q64_tree_select(const uint64_t (&t)[n], uint32_t q) {
    if constexpr (n == 1) {
        return t[0];
    } else {
        constexpr int m = (n + 1) / 2;
        uint64_t u[m];
        const uint64_t mb = 0ull - (uint64_t)(q & 1u);  // arithmetic blend (see kernels_qm.hip tree_select)
#pragma unroll
        for (int k = 0; k < m; ++k) u[k] = (2 * k + 1 < n) ? ((t[2 * k + 1] & mb) | (t[2 * k] & ~mb)) : t[2 * k];
        return q64_tree_select<m>(u, q >> 1);
    }
}

template <int NS>
__device__ inline void q64_load(const uint4 *tile, uint32_t lane, Q64Rows<NS> &s) {
#pragma unroll
    for (int g = 0; g < NS / 2; ++g) {
        const uint4 v = tile[g * 64 + lane];
        s.r[2 * g] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        s.r[2 * g + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
}
template <int NS>
__device__ inline void q64_store_group(uint4 *tile, uint32_t lane, const Q64Rows<NS> &s, int g) {
    tile[g * 64 + lane] = make_uint4((uint32_t)s.r[2 * g], (uint32_t)(s.r[2 * g] >> 32), (uint32_t)s.r[2 * g + 1], (uint32_t)(s.r[2 * g + 1] >> 32));
}

// identity word expected in `slot` when the env is solved (0 for unused slots)
template <int NS, bool HAS_Z>
__device__ inline uint64_t q64_identity_word(int slot, uint32_t N) {
    if (HAS_Z) {
        const uint32_t j = (uint32_t)slot >> 1;
        if (j >= N) return 0ull;
        return (slot & 1) ? (1ull << N) << j : 1ull << j;
    }
    return (uint32_t)slot < N ? 1ull << slot : 0ull;
}

template <int NS, bool HAS_Z>
__device__ inline bool q64_solved(const Q64Rows<NS> &s, uint32_t N) {  // clifford.rs:136-145
    uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i & 3] |= s.r[i] ^ q64_identity_word<NS, HAS_Z>(i, N);
    return ((acc[0] | acc[1]) | (acc[2] | acc[3])) == 0;
}

// bit j: qubit j's rows (CliffordEnv) / row j (LinearFunctionEnv) differ from the identity's
template <int NS, bool HAS_Z>
__device__ inline uint64_t q64_badmask(const Q64Rows<NS> &s, uint32_t N) {
    uint64_t bad = 0;
    if constexpr (HAS_Z) {
#pragma unroll
        for (int j = 0; j < NS / 2; ++j)
            bad |= (uint64_t)(s.r[2 * j] != q64_identity_word<NS, true>(2 * j, N) || s.r[2 * j + 1] != q64_identity_word<NS, true>(2 * j + 1, N)) << j;
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) bad |= (uint64_t)(s.r[j] != q64_identity_word<NS, false>(j, N)) << j;
    }
    return bad;
}

// apply one action; returns the mask of 16-byte groups written
template <int NS, bool HAS_Z>
__device__ inline uint64_t q64_apply(Q64Rows<NS> &s, uint32_t ops) {
    constexpr int NX = HAS_Z ? NS / 2 : NS;  // number of X-type rows
    const uint32_t q0 = ops & 63u, q1 = (ops >> 6) & 63u, m = (ops >> 12) & 0xFFFFu;
    uint64_t xs[NX], zs[HAS_Z ? NX : 1];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        xs[j] = s.r[HAS_Z ? 2 * j : j];
        if (HAS_Z) zs[j] = s.r[2 * j + 1];
    }
    const uint64_t x0 = q64_tree_select<NX>(xs, q0), x1 = q64_tree_select<NX>(xs, q1);
    uint64_t z0 = 0, z1 = 0;
    if constexpr (HAS_Z) {
        z0 = q64_tree_select<NX>(zs, q0);
        z1 = q64_tree_select<NX>(zs, q1);
    }
    auto mix = [&](uint32_t k) -> uint64_t {
        const uint32_t b = m >> (4 * k);
        uint64_t o = ((0ull - (uint64_t)(b & 1u)) & x0) ^ ((0ull - (uint64_t)((b >> 2) & 1u)) & x1);
        if (HAS_Z) o ^= ((0ull - (uint64_t)((b >> 1) & 1u)) & z0) ^ ((0ull - (uint64_t)((b >> 3) & 1u)) & z1);
        return o;
    };
    const uint64_t nx0 = mix(0), nx1 = mix(2);
    const uint64_t nz0 = HAS_Z ? mix(1) : 0ull, nz1 = HAS_Z ? mix(3) : 0ull;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const bool h0 = q0 == (uint32_t)j, h1 = q1 == (uint32_t)j;
        uint64_t vx = s.r[HAS_Z ? 2 * j : j];
        vx = h1 ? nx1 : vx;  // flat selects; q0's value wins when q0 == q1
        vx = h0 ? nx0 : vx;
        s.r[HAS_Z ? 2 * j : j] = vx;
        if (HAS_Z) {
            uint64_t vz = s.r[2 * j + 1];
            vz = h1 ? nz1 : vz;
            vz = h0 ? nz0 : vz;
            s.r[2 * j + 1] = vz;
        }
    }
    const uint32_t gsh = HAS_Z ? 0u : 1u;  // CliffordEnv: group = qubit; LinearFunctionEnv: group = 2 rows
    const uint64_t dirty = (1ull << (q0 >> gsh)) | (1ull << (q1 >> gsh));
    return m == Q64_IDENTITY ? 0ull : dirty;
}

// ---- add_inverts for CliffordEnv (see the discussion in kernels_qm.hip) -------------------------
// Slot space: a square R x R matrix, R = NS = 2*NQ, rows X-type first then Z-type, logical Z
// columns N..2N-1 moved to bit positions NQ..NQ+N-1.
__device__ inline uint64_t q64_cols_to_slots(uint64_t w, uint32_t N, uint32_t nq) {
    const uint64_t xm = (1ull << N) - 1ull;
    return (w & xm) | (((w >> N) & xm) << nq);
}
__device__ inline uint64_t q64_cols_from_slots(uint64_t w, uint32_t N, uint32_t nq) {
    const uint64_t xm = (1ull << N) - 1ull;
    return (w & xm) | (((w >> nq) & xm) << N);
}
__device__ inline void q64_transpose64(uint64_t (&a)[64]) {
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int j = 32 >> st;
        const uint64_t m = st == 0 ? 0x00000000FFFFFFFFull : st == 1 ? 0x0000FFFF0000FFFFull : st == 2 ? 0x00FF00FF00FF00FFull
                           : st == 3 ? 0x0F0F0F0F0F0F0F0Full : st == 4 ? 0x3333333333333333ull : 0x5555555555555555ull;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if ((k & j) == 0) {
                const uint64_t t = ((a[k] >> j) ^ a[k + j]) & m;
                a[k + j] ^= t;
                a[k] ^= t << j;
            }
        }
    }
}
template <int NS>
__device__ inline void q64_to_slot_space(const Q64Rows<NS> &s, uint32_t N, uint64_t (&m)[64]) {
    constexpr int NQ = NS / 2;
#pragma unroll
    for (int i = 0; i < 64; ++i)
        m[i] = i < NQ ? q64_cols_to_slots(s.r[2 * i], N, NQ) : (i < NS ? q64_cols_to_slots(s.r[2 * (i - NQ) + 1], N, NQ) : 0ull);
}
template <int NS>
__device__ inline void q64_from_slot_space(Q64Rows<NS> &s, uint32_t N, const uint64_t (&m)[64]) {
    constexpr int NQ = NS / 2;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        s.r[2 * i] = q64_cols_from_slots(m[i], N, NQ);
        s.r[2 * i + 1] = q64_cols_from_slots(m[NQ + i], N, NQ);
    }
}
// Omega M^T Omega in slot space
template <int NS>
__device__ inline void q64_symplectic_candidate(const uint64_t (&m)[64], uint64_t (&c)[64]) {
    constexpr int NQ = NS / 2;
    uint64_t t[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) t[i] = m[i];
    q64_transpose64(t);
    const uint64_t rmask = NS == 64 ? ~0ull : ((1ull << NS) - 1ull);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (i < NS) {
            const uint64_t w = t[(i + NQ) % NS];
            c[i] = ((w >> NQ) | (w << NQ)) & rmask;
        } else {
            c[i] = 0;
        }
    }
}
template <int NS>
__device__ inline bool q64_is_inverse(const uint64_t (&m)[64], const uint64_t (&c)[64], uint32_t N) {
    constexpr int NQ = NS / 2;
    uint64_t bad = 0;
#pragma unroll 1
    for (int i = 0; i < NS; ++i) {
        uint64_t row = 0;
#pragma unroll
        for (int k = 0; k < 64; ++k) row = (k == i) ? m[k] : row;
        uint64_t acc = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) acc ^= (0ull - ((row >> j) & 1ull)) & c[j];
        const bool real = (uint32_t)(i % NQ) < N;
        bad |= acc ^ (real ? 1ull << i : 0ull);
    }
    return bad == 0;
}
// general Gauss-Jordan (row operations only) for envs not known to be symplectic
template <int NS>
__device__ __noinline__ bool q64_gauss_jordan(Q64Rows<NS> &s, uint32_t N) {
    constexpr int NQ = NS / 2;
    uint64_t m[64], v[64];
    q64_to_slot_space<NS>(s, N, m);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const bool pad = i < NS && (uint32_t)(i % NQ) >= N;
        if (pad) m[i] = 1ull << i;
        v[i] = i < NS ? 1ull << i : 0ull;
    }
    uint64_t singular = 0;
#pragma unroll 1
    for (int col = 0; col < NS; ++col) {
        uint64_t pm = 0, pv = 0;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            pm = (k == col) ? m[k] : pm;
            pv = (k == col) ? v[k] : pv;
        }
        uint64_t need = ((pm >> col) & 1ull) - 1ull;
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const uint64_t t = (r > col && r < NS) ? (need & (0ull - ((m[r] >> col) & 1ull))) : 0ull;
            pm ^= m[r] & t;
            pv ^= v[r] & t;
            need &= ~t;
        }
        singular |= need;
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const uint64_t t = (r != col && r < NS) ? (0ull - ((m[r] >> col) & 1ull)) : 0ull;
            m[r] = (r == col) ? pm : (m[r] ^ (pm & t));
            v[r] = (r == col) ? pv : (v[r] ^ (pv & t));
        }
    }
    if (singular) return false;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const bool pad = i < NS && (uint32_t)(i % NQ) >= N;
        if (pad) v[i] = 0;
    }
    q64_from_slot_space<NS>(s, N, v);
    return true;
}
// in-place form of Omega M^T Omega for the step kernel: one 64-register work array
template <int NS>
__device__ inline void q64_symplectic_inverse(Q64Rows<NS> &s, uint32_t N) {
    constexpr int NQ = NS / 2;
    uint64_t m[64];
    q64_to_slot_space<NS>(s, N, m);
    q64_transpose64(m);
    const uint64_t rmask = NS == 64 ? ~0ull : ((1ull << NS) - 1ull);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const uint64_t wx = m[i + NQ], wz = m[i];  // slot-space rows (i + NQ) % NS and (i + 2 NQ) % NS
        s.r[2 * i] = q64_cols_from_slots(((wx >> NQ) | (wx << NQ)) & rmask, N, NQ);
        s.r[2 * i + 1] = q64_cols_from_slots(((wz >> NQ) | (wz << NQ)) & rmask, N, NQ);
    }
}
template <int NS>
__device__ __noinline__ bool q64_check_symplectic(const Q64Rows<NS> &s, uint32_t N) {
    uint64_t m[64], c[64];
    q64_to_slot_space<NS>(s, N, m);
    q64_symplectic_candidate<NS>(m, c);
    return q64_is_inverse<NS>(m, c, N);
}

// EXTRA: solution log / layer metrics / multi-step; INV: add_inverts (CliffordEnv); GJ: also compile
// the general Gauss-Jordan inversion (needed only when some env holds a non-symplectic matrix)
template <int NS, bool HAS_Z, bool EXTRA, bool INV, bool GJ = false>
__global__ __launch_bounds__(256) void q64_step_kernel(StepArgs a) {
    using Rows = Q64Rows<NS>;
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    if (env >= a.B) return;
    const bool act64 = a.flags & F_ACT64;
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64);
    int64_t act = load_action(a.actions, env, act64);
    Rows s;
    q64_load<NS>(tile, lane, s);
    int32_t depth = a.depth[env];
    uint32_t iflags = INV ? a.inverted[env] : 0u;
    uint64_t dirty = 0;
    bool solved = false;
    float reward = 0.0f;
    uint32_t fault = 0;
    int32_t sol_n = (EXTRA && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    int32_t sol_b = (EXTRA && INV && (a.flags & F_TRACK)) ? a.sol_len[env * 2 + 1] : 0;

    const uint32_t T = EXTRA ? a.T : 1u;
    for (uint32_t t = 0; t < T; ++t) {
        if (EXTRA && t) act = load_action(a.actions, (uint64_t)t * a.B + env, act64);
        const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // clifford.rs:324
        GateEntry g = {Q64_IDENTITY << 12, 0.0f};
        if (in_range) g = a.gates[act];
        float penalty = g.penalty;
        if (EXTRA && (a.flags & F_LAYERS) && in_range) penalty = layers_penalty(layer_rec(a.layers, env, 2 * a.N + 2), a.N, a.descs[act], a.w);
        dirty |= q64_apply<NS, HAS_Z>(s, g.ops);  // clifford.rs:331
        if (EXTRA && (a.flags & F_TRACK)) {  // clifford.rs:334-340
            if ((uint32_t)(sol_n + sol_b) < a.sol_cap) {
                const bool inv_frame = INV && (iflags & Q64_FLAG_INVERTED);
                sol_at(a, env, (uint32_t)(sol_n + sol_b)) = sol_word_framed(act, inv_frame);
                if (inv_frame) ++sol_b;
                else ++sol_n;
            } else {
                fault |= 8u;
            }
        }
        depth = depth > 0 ? depth - 1 : 0;  // clifford.rs:342
        if constexpr (INV && HAS_Z) {        // maybe_random_invert (clifford.rs:262-270)
            const uint32_t coin = a.coins ? a.coins[(uint64_t)t * a.B + env]
                                          : (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a) + t) >> 63);
            if (coin & 1u) {
                if (iflags & Q64_FLAG_SYMPLECTIC) {
                    q64_symplectic_inverse<NS>(s, a.N);
                    iflags ^= Q64_FLAG_INVERTED;
                    dirty = ~0ull;
                } else if constexpr (GJ) {
                    if (q64_gauss_jordan<NS>(s, a.N)) {
                        iflags ^= Q64_FLAG_INVERTED;
                        dirty = ~0ull;
                    } else {
                        fault |= QG_FAULT_SINGULAR;
                    }
                } else {
                    fault |= QG_FAULT_BAD_STATE;  // unreachable: see kernels_qm.hip
                }
            }
        }
        solved = q64_solved<NS, HAS_Z>(s, a.N);  // clifford.rs:344
        const float achieved = solved ? 1.0f : 0.0f;
        reward = achieved - penalty;             // clifford.rs:345-346
        if (EXTRA && a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
        if (EXTRA && a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
    }
#pragma unroll
    for (int g = 0; g < Rows::G; ++g)
        if ((dirty >> g) & 1ull) q64_store_group<NS>(tile, lane, s, g);
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (EXTRA && (a.flags & F_TRACK)) {
        a.sol_len[env * 2] = sol_n;
        if (INV) a.sol_len[env * 2 + 1] = sol_b;
    }
    if (INV) a.inverted[env] = (uint8_t)iflags;
    if ((EXTRA || INV) && fault) atomicOr(&a.error[env], fault);
    if (a.bad) reinterpret_cast<uint64_t *>(a.bad)[env] = q64_badmask<NS, HAS_Z>(s, a.N);  // the one-step kernel may run next
}

// ------------------------------------------------------------------------------------------
// The reference-default step (add_inverts, one step per launch, every env symplectic) with TWO lanes per env, as qm_inv2_kernel does
// for N <= 16: q64_step_kernel above is ~5 800 instructions per env at one wave per SIMD (64 x 64 transpose in 64-bit words, the
// column shuffles into and out of "slot space", select trees over 64 rows).  Here lane h of a pair holds the X-type (h = 0) or
// Z-type (h = 1) rows of all qubits, each as two 32-bit words {lo = X columns, hi = Z columns}: the matrix is four 32 x 32 blocks
// [A B; C D] (lane 0: A | B, lane 1: C | D), always padded to 32 qubits with identity rows (the inverse of diag(M, 1) is
// diag(M^-1, 1)), so that
//     M^-1 = Omega M^T Omega:   X row i = {lo = D^T_i, hi = B^T_i},   Z row i = {lo = C^T_i, hi = A^T_i}
// is two 32 x 32 transposes in each lane's own registers (32-bit operations) and ONE exchange of 32 words with the partner lane
// (DPP quad_perm [1, 0, 3, 2]); the gate's 4 x 4 GF(2) map needs the partner's two rows (4 DPP moves).
// ------------------------------------------------------------------------------------------
__device__ inline uint32_t q64_pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }

template <int n>
__device__ inline uint32_t q64_tree_select32(const uint32_t (&t)[n], uint32_t q) {
    if constexpr (n == 1) {
        return t[0];
    } else {
        constexpr int m = (n + 1) / 2;
        uint32_t u[m];
        const uint32_t mb = 0u - (q & 1u);
#pragma unroll
        for (int k = 0; k < m; ++k) u[k] = (2 * k + 1 < n) ? ((t[2 * k + 1] & mb) | (t[2 * k] & ~mb)) : t[2 * k];
        return q64_tree_select32<m>(u, q >> 1);
    }
}

__device__ inline void q64_transpose32(uint32_t (&w)[32]) {
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int d = 16 >> st;
        const uint32_t m = st == 0 ? 0x0000FFFFu : st == 1 ? 0x00FF00FFu : st == 2 ? 0x0F0F0F0Fu : st == 3 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if ((i & d) == 0) {
                const uint32_t lo = w[i], hi = w[i + d];
                w[i] = (lo & m) | ((hi & m) << d);
                w[i + d] = ((lo >> d) & m) | (hi & ~m);
            }
        }
    }
}

// NQ = qubit slots in memory (NS / 2 = N rounded up to 4)
template <int NQ, bool FEAT>
__global__ __launch_bounds__(256) void q64_inv2_kernel(StepArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = tid >> 1;
    const uint32_t h = (uint32_t)tid & 1u;
    QG_PREFETCH_STEP_ARGS(a);
    if (env >= a.B) return;  // whole lane pairs leave together
    const uint32_t N = a.N;
    // group q = {X row of qubit q, Z row of qubit q} (two uint64): this lane's half of it
    uint2 *rows = reinterpret_cast<uint2 *>(reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(NQ * 64) + (env & 63u)) + h;
    uint2 raw[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) raw[q] = rows[(uint32_t)q * 128u];  // 64 uint4 = 128 uint2 per group
    int32_t depth = a.depth[env];
    uint32_t iflags = a.inverted[env];
    uint32_t coin = a.coins ? a.coins[env] : 0u;
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    int32_t sol_b = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2 + 1] : 0;
    const int64_t act = load_action(a.actions, env, a.flags & F_ACT64);
    const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
    GateEntry g = a.gates[in_range ? act : 0];  // unconditional (clamped) load: nothing else waits behind a branch
    if (!in_range) g = GateEntry{Q64_IDENTITY << 12, 0.0f};
    if (!a.coins) coin = (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a)) >> 63);  // runs under the loads
    // block form: stored word = X columns | Z columns << N; rows of the padding qubits N .. 31 are the identity's
    uint32_t lo[32], hi[32];
    const uint32_t xm = N >= 32 ? 0xFFFFFFFFu : ((1u << N) - 1u);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        uint32_t l = h ? 0u : 1u << q, u = h ? 1u << q : 0u;
        if (q < NQ) {
            const uint64_t w = (uint64_t)raw[q].x | ((uint64_t)raw[q].y << 32);
            const bool real = (uint32_t)q < N;
            l = real ? (uint32_t)w & xm : l;
            u = real ? (uint32_t)(w >> N) & xm : u;
        }
        lo[q] = l;
        hi[q] = u;
    }
    uint32_t fault = 0;
    float penalty = g.penalty;
    if (FEAT && (a.flags & F_LAYERS) && in_range && h == 0) penalty = layers_penalty(layer_rec(a.layers, env, 2 * N + 2), N, a.descs[act], a.w);

    // ---- apply_gate_to_state (clifford.rs:331): the 4x4 GF(2) map on {X[q0], Z[q0], X[q1], Z[q1]}; this lane makes its own type's two rows ----
    uint32_t dirty = 0;  // qubits whose rows changed
    {
        const uint32_t q0 = g.ops & 63u, q1 = (g.ops >> 6) & 63u, m = (g.ops >> 12) & 0xFFFFu;
        const uint32_t o0l = q64_tree_select32<32>(lo, q0), o0h = q64_tree_select32<32>(hi, q0);
        const uint32_t o1l = q64_tree_select32<32>(lo, q1), o1h = q64_tree_select32<32>(hi, q1);
        const uint32_t p0l = q64_pair_swap(o0l), p0h = q64_pair_swap(o0h), p1l = q64_pair_swap(o1l), p1h = q64_pair_swap(o1h);
        // inputs in the order X0, Z0, X1, Z1: the X rows are lane 0's
        const uint32_t x0l = h ? p0l : o0l, x0h = h ? p0h : o0h, z0l = h ? o0l : p0l, z0h = h ? o0h : p0h;
        const uint32_t x1l = h ? p1l : o1l, x1h = h ? p1h : o1h, z1l = h ? o1l : p1l, z1h = h ? o1h : p1h;
        auto mix = [&](uint32_t k, uint32_t &ol, uint32_t &oh) {  // out_k = xor_i M[k][i] * in_i
            const uint32_t b = m >> (4 * k);
            const uint32_t s0 = 0u - (b & 1u), s1 = 0u - ((b >> 1) & 1u), s2 = 0u - ((b >> 2) & 1u), s3 = 0u - ((b >> 3) & 1u);
            ol = (s0 & x0l) ^ (s1 & z0l) ^ (s2 & x1l) ^ (s3 & z1l);
            oh = (s0 & x0h) ^ (s1 & z0h) ^ (s2 & x1h) ^ (s3 & z1h);
        };
        uint32_t n0l, n0h, n1l, n1h;
        mix(h, n0l, n0h);       // X[q0] (k = 0) or Z[q0] (k = 1)
        mix(2u + h, n1l, n1h);  // X[q1] (k = 2) or Z[q1] (k = 3)
        if (m != Q64_IDENTITY) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {  // q1's rows first, then q0's (q0's value wins when q0 == q1, as in q64_apply)
                const bool h1 = q1 == (uint32_t)j, h0 = q0 == (uint32_t)j;
                uint32_t vl = lo[j], vh = hi[j];
                vl = h1 ? n1l : vl; vh = h1 ? n1h : vh;
                vl = h0 ? n0l : vl; vh = h0 ? n0h : vh;
                lo[j] = vl; hi[j] = vh;
            }
            dirty = (1u << q0) | (1u << q1);
        }
    }

    if (FEAT && (a.flags & F_TRACK) && h == 0) {  // clifford.rs:334-340: entries in push order, bit 31 = pushed to solution_inv
        if ((uint32_t)(sol_n + sol_b) < a.sol_cap) {
            const bool inv_frame = iflags & Q64_FLAG_INVERTED;
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
        if (iflags & Q64_FLAG_SYMPLECTIC) {
            q64_transpose32(lo);  // lane 0: A^T, lane 1: C^T
            q64_transpose32(hi);  // lane 0: B^T, lane 1: D^T
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const uint32_t got = q64_pair_swap(h ? hi[i] : lo[i]);  // lane 0 receives D^T, lane 1 receives A^T
                const uint32_t nl = h ? lo[i] : got, nh = h ? got : hi[i];
                lo[i] = nl;
                hi[i] = nh;
            }
            iflags ^= Q64_FLAG_INVERTED;
            dirty = 0xFFFFFFFFu;
        } else {
            fault |= QG_FAULT_BAD_STATE;  // unreachable: the host launches the Gauss-Jordan variant whenever such an env may exist
        }
    }

    // ---- solved (clifford.rs:344), the per-qubit mask the one-step kernel keeps, reward (:345-346) --------------------------------
    uint32_t bad = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool real = (uint32_t)q < N;
        const uint32_t wl = h ? 0u : 1u << q, wh = h ? 1u << q : 0u;
        bad |= (uint32_t)(real && (lo[q] != wl || hi[q] != wh)) << q;
    }
    bad |= q64_pair_swap(bad);
    const bool solved = bad == 0;
    const float achieved = solved ? 1.0f : 0.0f;
    const float reward = achieved - penalty;

#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (((dirty >> q) & 1u) && (uint32_t)q < N) {
            const uint64_t w = (uint64_t)lo[q] | ((uint64_t)hi[q] << N);
            rows[(uint32_t)q * 128u] = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
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
        if (a.bad) reinterpret_cast<uint64_t *>(a.bad)[env] = (uint64_t)bad;  // the one-step kernel may run next
    }
    if (a.flags & F_DONE_LIST) done_mask_store_pairs(a.done_mask, a.B, h == 0 && (depth == 0 || solved), tid, a.done_epoch);  // (every lane still alive)
}

// One step per launch without holding the matrix (see qm_step1_kernel in kernels_qm.hip): the gate's
// <= 2 groups ({X[q], Z[q]} of a qubit for CliffordEnv, a row pair for LinearFunctionEnv) are gathered
// and scattered at per-lane addresses, `solved` comes from the incrementally kept 64-bit `bad` mask.
// LIST: also append the envs that finish to StepArgs::done_list (F_DONE_LIST; its own instantiation: the plain kernel's code stays as it is)
// (`act`, and `g` = gates[act] when it is in range: loaded by the caller -- q64_reset_step_kernel's reset lanes ask for them before the scramble, the env's first step
// then has one trip to memory left, not three)
template <int NS, bool HAS_Z, bool FEAT>
__device__ __forceinline__ bool q64_step1_with(const StepArgs &a, uint64_t env, int64_t act, GateEntry g) {  // returns is_final
    const uint32_t lane = (uint32_t)(env & (QG_WAVE - 1));  // (= the thread's lane in the step kernels; a reset's lane steps the env it has just written from wherever it sits)
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Q64Rows<NS>::G * 64);
    uint64_t *badp = reinterpret_cast<uint64_t *>(a.bad) + env;
    int32_t depth = a.depth[env];
    const uint64_t bad0 = *badp;
    uint64_t bad = bad0;
    uint32_t fault = 0;
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // clifford.rs:324
    float penalty = 0.0f;
    if (in_range) {
        penalty = g.penalty;
        if (FEAT && (a.flags & F_LAYERS)) penalty = layers_penalty(layer_rec(a.layers, env, 2 * a.N + 2), a.N, a.descs[act], a.w);
        const uint32_t q0 = g.ops & 63u, q1 = (g.ops >> 6) & 63u, m = (g.ops >> 12) & 0xFFFFu;
        if (m != Q64_IDENTITY) {
            constexpr uint32_t gsh = HAS_Z ? 0u : 1u;
            const uint32_t g0 = q0 >> gsh, g1 = q1 >> gsh;
            const uint4 va = tile[g0 * 64 + lane], vb = tile[g1 * 64 + lane];
            uint64_t ua[2] = {(uint64_t)va.x | ((uint64_t)va.y << 32), (uint64_t)va.z | ((uint64_t)va.w << 32)};
            uint64_t ub[2] = {(uint64_t)vb.x | ((uint64_t)vb.y << 32), (uint64_t)vb.z | ((uint64_t)vb.w << 32)};
            const uint64_t o0 = 0ull - (uint64_t)(q0 & 1u), o1 = 0ull - (uint64_t)(q1 & 1u);  // LinearFunctionEnv: odd row of the pair
            const uint64_t x0 = HAS_Z ? ua[0] : ((ua[1] & o0) | (ua[0] & ~o0)), z0 = HAS_Z ? ua[1] : 0ull;
            const uint64_t x1 = HAS_Z ? ub[0] : ((ub[1] & o1) | (ub[0] & ~o1)), z1 = HAS_Z ? ub[1] : 0ull;
            auto mix = [&](uint32_t k) -> uint64_t {
                const uint32_t b = m >> (4 * k);
                uint64_t o = ((0ull - (uint64_t)(b & 1u)) & x0) ^ ((0ull - (uint64_t)((b >> 2) & 1u)) & x1);
                if (HAS_Z) o ^= ((0ull - (uint64_t)((b >> 1) & 1u)) & z0) ^ ((0ull - (uint64_t)((b >> 3) & 1u)) & z1);
                return o;
            };
            const uint64_t nx0 = mix(0), nx1 = mix(2), nz0 = HAS_Z ? mix(1) : 0ull, nz1 = HAS_Z ? mix(3) : 0ull;
            // q1's rows first, then q0's (q0's value wins when q0 == q1, as in q64_apply)
            if constexpr (HAS_Z) {
                ub[0] = nx1; ub[1] = nz1;
            } else {
                ub[0] = (ub[0] & o1) | (nx1 & ~o1); ub[1] = (nx1 & o1) | (ub[1] & ~o1);
            }
            const bool same = g0 == g1;
            ua[0] = same ? ub[0] : ua[0];
            ua[1] = same ? ub[1] : ua[1];
            if constexpr (HAS_Z) {
                ua[0] = nx0; ua[1] = nz0;
            } else {
                ua[0] = (ua[0] & o0) | (nx0 & ~o0); ua[1] = (nx0 & o0) | (ua[1] & ~o0);
            }
            if (!same) tile[g1 * 64 + lane] = make_uint4((uint32_t)ub[0], (uint32_t)(ub[0] >> 32), (uint32_t)ub[1], (uint32_t)(ub[1] >> 32));
            tile[g0 * 64 + lane] = make_uint4((uint32_t)ua[0], (uint32_t)(ua[0] >> 32), (uint32_t)ua[1], (uint32_t)(ua[1] >> 32));
            const uint64_t zb = 1ull << a.N;
            const uint64_t b1 = (uint64_t)(nx1 != (1ull << q1) || (HAS_Z && nz1 != (zb << q1)));
            const uint64_t b0 = (uint64_t)(nx0 != (1ull << q0) || (HAS_Z && nz0 != (zb << q0)));
            bad = (bad & ~(1ull << q1)) | (b1 << q1);
            bad = (bad & ~(1ull << q0)) | (b0 << q0);
        }
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
    if (bad != bad0) *badp = bad;
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (FEAT && (a.flags & F_TRACK)) a.sol_len[env * 2] = sol_n;
    if (FEAT && fault) atomicOr(&a.error[env], fault);
    return depth == 0 || solved;
}
template <int NS, bool HAS_Z, bool FEAT>
__device__ __forceinline__ bool q64_step1_body(const StepArgs &a, uint64_t env) {  // returns is_final
    const int64_t act = load_action(a.actions, env, a.flags & F_ACT64);
    GateEntry g{0u, 0.0f};
    if (act >= 0 && act < (int64_t)a.num_actions) g = a.gates[act];
    return q64_step1_with<NS, HAS_Z, FEAT>(a, env, act, g);
}
// (LIST: the envs that finish are left as one bit each in StepArgs::done_mask -- the wave's ballot, device_common.hpp done_mask_store; round 4 appended
// their indices to a list with one atomic per workgroup of 1 024 threads)
template <int NS, bool HAS_Z, bool FEAT, bool LIST = false>
__global__ __launch_bounds__(256) void q64_step1_kernel(StepArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    QG_PREFETCH_STEP_ARGS(a);  // qgym_internal.hpp
    if constexpr (LIST) {  // every thread reaches the wave's ballot; qg_vec_reset_done follows (qgym_api.cpp)
        const bool fin = env < a.B && q64_step1_body<NS, HAS_Z, FEAT>(a, env);
        done_mask_store(a.done_mask, a.B, fin, env, a.done_epoch);
    } else {
        if (env >= a.B) return;
        (void)q64_step1_body<NS, HAS_Z, FEAT>(a, env);
    }
}

// the tail of set_state / reset for one env: rows to the tile, reset_internals (clifford.rs:272-283)
template <int NS, bool HAS_Z, bool RESET_ONLY = false>
__device__ inline void q64_init_finish(const InitArgs &a, uint64_t env, const Q64Rows<NS> &s) {
    using Rows = Q64Rows<NS>;
    const uint32_t lane = (uint32_t)(env & (QG_WAVE - 1));
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64);
    const bool solved = q64_solved<NS, HAS_Z>(s, a.N);
#pragma unroll
    for (int g = 0; g < Rows::G; ++g) q64_store_group<NS>(tile, lane, s, g);
    uint32_t symp = 0;
    if constexpr (HAS_Z) {
        if (a.check_symplectic) {
            bool is_symp = true;  // (identity + gates: symplectic; set_state's matrix is checked -- 64 x 64 transposes in registers, kept out of the reset-only instantiation)
            if constexpr (!RESET_ONLY) is_symp = a.mode != 1 || q64_check_symplectic<NS>(s, a.N);
            symp = is_symp ? Q64_FLAG_SYMPLECTIC : 0u;
            if (!symp && a.nonsymp_flag) atomicOr(a.nonsymp_flag, 1u);
        }
    }
    if (a.bad) reinterpret_cast<uint64_t *>(a.bad)[env] = q64_badmask<NS, HAS_Z>(s, a.N);
    a.depth[env] = a.depth_value;  // reset_internals (clifford.rs:272-283)
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
    a.inverted[env] = (uint8_t)symp;
    a.error[env] = 0;
    a.sol_len[env * 2] = 0;
    a.sol_len[env * 2 + 1] = 0;
    if (a.layers) {
        const LayerRec lay = layer_rec(a.layers, env, a.layers_len);
        for (uint32_t i = 0; i + 2 < a.layers_len; ++i) lay[i] = -1;
        lay[a.layers_len - 2] = 0;
        lay[a.layers_len - 1] = 0;
    }
}

// The work of "workgroup" vblock (64 envs, or 4 with 16 lanes each) of an init launch.  Returns false -- on every thread alike -- when neither this vblock
// nor any later one has work (a list / mask of finished envs that ends before it, or that the trees have taken).
// T = 64: q64_init_kernel, a workgroup is one wave.  T = 256, RESET_ONLY: the workgroups behind the trees in q64_reset_done_kernel (qg_vec_reset_done in one launch; no
// set_state code, whose symplectic check alone takes 256 registers and scratch): all four waves count the mask (or take the list's length) and learn whether the
// trees of the same launch have taken the list -- then everybody leaves -- and three of them leave anyway before the first wave does the work.
// `after(env)`: called by the lane that has written env's fresh episode (state, depth, bad mask, log lengths) -- q64_reset_step_kernel's first step of that env.
// `after.ahead(env)`: called once the env is known, before its scramble, by the lane(s) that may finish it (that step's action and gate entry are requested there).
// WAVE_STEP (q64_reset_step_kernel without layer weights): the tree's wave takes the env's first step itself on the rows it holds across its lanes,
// before it stores them -- nothing the reset wrote is read back (reading back: +3 us a tree: the stores' round trip, an L1 invalidate, the loads).
struct Q64NoAfter {
    static constexpr bool WAVE_STEP = false;
    __device__ void ahead(uint64_t) {}
    __device__ void operator()(uint64_t) const {}
};
template <int NS, bool HAS_Z, uint32_t T = 64, bool RESET_ONLY = false, typename After = Q64NoAfter>
__device__ __forceinline__ bool q64_init_body(const InitArgs &a, uint32_t vblock, After after = After()) {
    using Rows = Q64Rows<NS>;
    // reset scramble on LDS-resident rows (device_common.hpp); one wave per block: NS * 512 B <= 32 KiB
    __shared__ uint64_t lds_rows[NS][QG_WAVE];
    const uint64_t tid = (uint64_t)vblock * QG_WAVE + threadIdx.x;
    uint64_t env = tid;
    Rows s;
    const bool trees = T != 64 && a.coop && a.n_draws >= 64u && a.tree_grid;  // this launch's first workgroups are trees and take the lists tree_takes says
    if (a.list) {  // qg_vec_reset_done: the finished envs are the bits of the mask the step before left (if it did), then the entries of the compacted / appended list
        __shared__ uint32_t mask_part[T + 2 + T / 64];
        DoneMaskShare share;
        uint32_t mcount = 0;
        if (a.mask) {
            done_mask_load<T>(a.mask, a.B, a.mask_words, share);
            mcount = *done_mask_hint(a.mask, a.B) != a.mask_epoch ? 0u : done_mask_scan<T>(share, mask_part);
        }
        const uint32_t len = mcount + a.list_count[0];
        const uint32_t lanes = (a.coop && coop_takes(len, a.B)) ? QG_COOP_LANES : 1u;
        // (T = 256: that launch reads its lists without tickets -- the host gives it InitArgs::zero_count -- and `threads` only says which workgroups have work)
        const uint32_t count = list_count_take(a.list_count, len, (uint64_t)len * lanes * (T / QG_WAVE), vblock, a.zero_count);
        // (without reader tickets -- InitArgs::zero_count -- the list the trees have taken is still there: the same test says so)
        if (a.zero_count && trees && tree_takes(count, a.n_draws)) return false;
        if (T != 64 && threadIdx.x >= QG_WAVE) return false;  // (after the scan's and list_count_take's barriers: nothing below has one)
        if ((uint64_t)vblock * QG_WAVE >= (uint64_t)count * lanes) return false;  // (the list ends before this vblock)
        const uint64_t *mask = a.mask;
        const uint32_t *list = a.list;
        const uint32_t words = a.mask_words;
        auto entry = [=](uint32_t i) -> uint32_t { return i < mcount ? done_mask_nth<T>(mask, words, mask_part, i) : list[i - mcount]; };
        if (lanes != 1u) {  // few finished envs: 16 lanes each
            const uint32_t N = a.N;
            if (tid / QG_COOP_LANES < count) after.ahead(entry((uint32_t)(tid / QG_COOP_LANES)));
            const uint64_t *rows = scramble_coop<uint64_t, NS>(a, count, &lds_rows[0][0], env, [N](uint32_t k) -> uint64_t { return q64_identity_word<NS, HAS_Z>((int)k, N); },
                                                               vblock, entry, QG_WAVE);
            if (!rows) return true;
#pragma unroll
            for (int k = 0; k < NS; ++k) s.r[k] = rows[k];
            q64_init_finish<NS, HAS_Z, RESET_ONLY>(a, env, s);
            after(env);
            return true;
        }
        if (tid >= count) return true;
        env = entry((uint32_t)tid);
        after.ahead(env);
    } else {
        if (RESET_ONLY && T != 64) return false;  // (behind the trees there is always a list; T = 64, RESET_ONLY: qg_vec_reset of the whole batch, or by the flags)
        if (env >= a.B) return true;
        if (a.only_done && !a.done[env]) return true;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) s.r[i] = q64_identity_word<NS, HAS_Z>(i, a.N);
    if (!RESET_ONLY && a.mode == 1) {  // set_state (clifford.rs:299-304)
#pragma unroll 1
        for (int sl = 0; sl < NS; ++sl) {
            const uint32_t j = HAS_Z ? (uint32_t)sl >> 1 : (uint32_t)sl;
            const uint32_t row = (HAS_Z && (sl & 1)) ? a.N + j : j;
            uint64_t w = 0;
            if (j < a.N) {
                if (a.format == QG_FMT_PACKED) {
                    w = reinterpret_cast<const uint64_t *>(a.src)[env * a.src_stride + row];
                    if (a.D < 64) w &= (1ull << a.D) - 1ull;
                } else if (a.format == QG_FMT_BITS) {  // the entry stream as bits (pack_bitstream): i64 / u8 set_state at streaming rate
                    w = bits_window(reinterpret_cast<const uint64_t *>(a.src), env * a.src_stride + (uint64_t)row * a.D, a.D);
                } else if (a.format == QG_FMT_I64) {
                    const int64_t *p = reinterpret_cast<const int64_t *>(a.src) + env * a.src_stride + (uint64_t)row * a.D;
                    for (uint32_t c = 0; c < a.D; ++c) w |= (uint64_t)(p[c] > 0) << c;
                } else {
                    const int8_t *p = reinterpret_cast<const int8_t *>(a.src) + env * a.src_stride + (uint64_t)row * a.D;
                    for (uint32_t c = 0; c < a.D; ++c) w |= (uint64_t)(p[c] > 0) << c;
                }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) s.r[k] = (k == sl) ? w : s.r[k];  // sl is wave-uniform
        }
    } else if (a.mode == 2) {  // reset scramble (clifford.rs:306-316)
        const uint32_t L = threadIdx.x & (QG_WAVE - 1);
#pragma unroll
        for (int k = 0; k < NS; ++k) lds_rows[k][L] = s.r[k];
        scramble_flat<uint64_t>(lds_rows, L, a, env);
#pragma unroll
        for (int k = 0; k < NS; ++k) s.r[k] = lds_rows[k][L];
    }
    q64_init_finish<NS, HAS_Z, RESET_ONLY>(a, env, s);
    after(env);
    return true;
}
// RESET_ONLY: mode 2 (qg_vec_reset, qg_vec_reset_done) -- without set_state's code, whose symplectic check makes this kernel 256 VGPRs and 400 bytes of scratch
template <int NS, bool HAS_Z, bool RESET_ONLY = false>
__global__ __launch_bounds__(64) void q64_init_kernel(InitArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    (void)q64_init_body<NS, HAS_Z, 64, RESET_ONLY>(a, blockIdx.x);
}

// export: one thread per (env, matrix row)
__global__ __launch_bounds__(256) void q64_export_kernel(ObsArgs a, uint32_t ns, uint32_t has_z) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid / a.D;
    const uint32_t row = (uint32_t)(gid % a.D);
    if (env >= a.B) return;
    const uint32_t slot = has_z ? (row < a.N ? 2 * row : 2 * (row - a.N) + 1) : row;
    const uint64_t *tile = reinterpret_cast<const uint64_t *>(a.state) + (env >> 6) * (uint64_t)(ns * 64);
    const uint64_t w = tile[((slot >> 1) * 64 + (uint32_t)(env & 63)) * 2 + (slot & 1)];
    if (a.format == QG_FMT_PACKED) {
        reinterpret_cast<uint64_t *>(a.out)[env * a.out_stride + row] = w;
    } else if (a.format == QG_FMT_I64) {
        int64_t *o = reinterpret_cast<int64_t *>(a.out) + env * a.out_stride + (uint64_t)row * a.D;
        for (uint32_t c = 0; c < a.D; ++c) o[c] = (int64_t)((w >> c) & 1ull);
    } else {
        int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride + (uint64_t)row * a.D;
        for (uint32_t c = 0; c < a.D; ++c) o[c] = (int8_t)((w >> c) & 1ull);
    }
}

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

// Fused rollout on LDS-resident rows: the uint64 counterpart of qm_fused_lds_kernel (kernels_qm.hip) -- one wave per workgroup,
// rows as [slot][lane] (NS x 512 B <= 32 KiB), a gate = four dynamic-index 64-bit LDS reads, the 4x4 GF(2) mix, four writes; the
// incremental 64-bit `bad` mask gives `solved`; actions and gate entries two batches of four steps ahead, all loads unconditional.
template <int NS, bool HAS_Z, bool ACT64>
__global__ __launch_bounds__(64) void q64_fused_lds_kernel(StepArgs a) {
    __shared__ uint64_t rows[NS][QG_WAVE];
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    if (env >= a.B) return;  // no cross-lane operation and no barrier below
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Q64Rows<NS>::G * 64);
    uint64_t bad;
    {
        Q64Rows<NS> s;
        q64_load<NS>(tile, lane, s);
        bad = q64_badmask<NS, HAS_Z>(s, a.N);
#pragma unroll
        for (int k = 0; k < NS; ++k) rows[k][lane] = s.r[k];
    }
    int32_t depth = a.depth[env];
    const uint64_t zb = 1ull << a.N;
    float reward = 0.0f;
    bool solved = bad == 0;
    auto load_act = [&](uint32_t t) -> int64_t {
        const uint64_t i = (uint64_t)(t < a.T ? t : a.T - 1u) * a.B + env;
        const int64_t v = ACT64 ? reinterpret_cast<const int64_t *>(a.actions)[i] : (int64_t) reinterpret_cast<const int32_t *>(a.actions)[i];
        return t < a.T ? v : -1;
    };
    auto load_gate = [&](int64_t act, GateEntry &g) {
        const bool ok = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
        const GateEntry e = a.gates[ok ? act : 0];
        g.ops = ok ? e.ops : (Q64_IDENTITY << 12);
        g.penalty = ok ? e.penalty : 0.0f;
    };
    auto step = [&](uint32_t t, const GateEntry &g) {
        if (t >= a.T) return;
        const uint32_t q0 = g.ops & 63u, q1 = (g.ops >> 6) & 63u, m = (g.ops >> 12) & 0xFFFFu;
        if (m != Q64_IDENTITY) {
            const uint32_t sx0 = HAS_Z ? 2u * q0 : q0, sx1 = HAS_Z ? 2u * q1 : q1;
            const uint64_t x0 = rows[sx0][lane], x1 = rows[sx1][lane];
            const uint64_t z0 = HAS_Z ? rows[sx0 + 1u][lane] : 0ull, z1 = HAS_Z ? rows[sx1 + 1u][lane] : 0ull;
            auto mix = [&](uint32_t k) -> uint64_t {
                const uint32_t b = m >> (4 * k);
                uint64_t o = ((0ull - (uint64_t)(b & 1u)) & x0) ^ ((0ull - (uint64_t)((b >> 2) & 1u)) & x1);
                if (HAS_Z) o ^= ((0ull - (uint64_t)((b >> 1) & 1u)) & z0) ^ ((0ull - (uint64_t)((b >> 3) & 1u)) & z1);
                return o;
            };
            const uint64_t nx0 = mix(0), nx1 = mix(2), nz0 = HAS_Z ? mix(1) : 0ull, nz1 = HAS_Z ? mix(3) : 0ull;
            // q1's rows first, then q0's (q0's value wins when q0 == q1, as in q64_apply)
            rows[sx1][lane] = nx1;
            if (HAS_Z) rows[sx1 + 1u][lane] = nz1;
            rows[sx0][lane] = nx0;
            if (HAS_Z) rows[sx0 + 1u][lane] = nz0;
            const uint64_t b1 = (uint64_t)(nx1 != (1ull << q1) || (HAS_Z && nz1 != (zb << q1)));
            const uint64_t b0 = (uint64_t)(nx0 != (1ull << q0) || (HAS_Z && nz0 != (zb << q0)));
            bad = (bad & ~(1ull << q1)) | (b1 << q1);
            bad = (bad & ~(1ull << q0)) | (b0 << q0);
        }
        depth = depth > 0 ? depth - 1 : 0;  // clifford.rs:342
        solved = bad == 0;                   // clifford.rs:344
        const float achieved = solved ? 1.0f : 0.0f;
        reward = achieved - g.penalty;       // clifford.rs:345-346
        if (a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
        if (a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
    };
    GateEntry cur[4], nxt[4];
    int64_t acts[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) load_gate(load_act(k), cur[k]);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) acts[k] = load_act(4 + k);
    for (uint32_t t = 0; t < a.T; t += 4) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) load_gate(acts[k], nxt[k]);      // batch t + 4
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) acts[k] = load_act(t + 8 + k);   // batch t + 8
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) step(t + k, cur[k]);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) cur[k] = nxt[k];
    }
#pragma unroll
    for (int g = 0; g < Q64Rows<NS>::G; ++g) {
        const uint64_t r0 = rows[2 * g][lane], r1 = rows[2 * g + 1][lane];
        tile[g * 64 + lane] = make_uint4((uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32));
    }
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (a.bad) reinterpret_cast<uint64_t *>(a.bad)[env] = bad;
}

template <int NS, bool HAS_Z>
static hipError_t q64_launch_step(const StepArgs &a, hipStream_t s) {
    const dim3 grid(grid_for(a.B, 256)), block(256);
    const bool feat = a.flags & (F_TRACK | F_LAYERS), list = a.flags & F_DONE_LIST;
    const bool extra = feat || a.T != 1 || a.rewards_seq || a.dones_seq;
    switch (plan::tile64_step(a.flags, a.T, a.bad != nullptr, a.rewards_seq || a.dones_seq, a.num_actions, HAS_Z)) {  // qgym_plan.hpp
    case plan::SK_Q64_STEP1:  // the env.step() path
        if (list && !a.done_mask) return hipErrorInvalidValue;  // (qgym_api.cpp step_leaves_done_list gives every list-leaving step a mask)
        if (feat && list) hipLaunchKernelGGL((q64_step1_kernel<NS, HAS_Z, true, true>), grid, block, 0, s, a);
        else if (feat) hipLaunchKernelGGL((q64_step1_kernel<NS, HAS_Z, true>), grid, block, 0, s, a);
        else if (list) hipLaunchKernelGGL((q64_step1_kernel<NS, HAS_Z, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((q64_step1_kernel<NS, HAS_Z, false>), grid, block, 0, s, a);
        return hipGetLastError();
    case plan::SK_Q64_FUSED_LDS: {  // plain fused rollout: rows in LDS, one wave per workgroup
        const dim3 g1(grid_for(a.B, 64)), b1(64);
        if (a.flags & F_ACT64) hipLaunchKernelGGL((q64_fused_lds_kernel<NS, HAS_Z, true>), g1, b1, 0, s, a);
        else hipLaunchKernelGGL((q64_fused_lds_kernel<NS, HAS_Z, false>), g1, b1, 0, s, a);
        return hipGetLastError();
    }
    case plan::SK_Q64_INV2:  // every env symplectic, one step per launch: two lanes per env
        if constexpr (HAS_Z) {
            const dim3 grid2(grid_for(2 * a.B, 256));
            if (feat) hipLaunchKernelGGL((q64_inv2_kernel<NS / 2, true>), grid2, block, 0, s, a);
            else hipLaunchKernelGGL((q64_inv2_kernel<NS / 2, false>), grid2, block, 0, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    case plan::SK_Q64_STEP_GJ:
        if constexpr (HAS_Z) {
            hipLaunchKernelGGL((q64_step_kernel<NS, HAS_Z, true, true, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    case plan::SK_Q64_STEP_INV:
        if constexpr (HAS_Z) {
            hipLaunchKernelGGL((q64_step_kernel<NS, HAS_Z, true, true, false>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    case plan::SK_Q64_STEP:
        if (extra) hipLaunchKernelGGL((q64_step_kernel<NS, HAS_Z, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((q64_step_kernel<NS, HAS_Z, false, false>), grid, block, 0, s, a);
        return hipGetLastError();
    default: return hipErrorInvalidValue;
    }
}
// qg_vec_reset_done with a short list of long scrambles (tree_takes): a workgroup per listed env, scramble_tree64.  Runs BEFORE q64_init_kernel
// in the same call and consumes the list (the ticket of list_count_take zeroes it, or -- no tickets, InitArgs::zero_count -- the init kernel applies
// the same test), so that the init kernel finds nothing to do; when the list is too long or the scramble too short it leaves the list alone and the
// init kernel takes it as before.
// Q64_TREE_WAVES waves per workgroup, each with n / 4 gates of the chain, then two levels of products (eight waves measured: slower, device_common.hpp
// scramble_tree64_ops)
constexpr int Q64_TREE_WAVES = 4;
constexpr uint32_t Q64_TREE_THREADS = 64u * Q64_TREE_WAVES;
template <int NS, bool HAS_Z, typename After = Q64NoAfter>
__device__ __forceinline__ void q64_reset_tree_body(const InitArgs &a, After after = After()) {  // workgroups blockIdx.x < a.tree_grid of q64_reset_done_kernel / q64_reset_step_kernel
    __shared__ uint64_t prod[Q64_TREE_WAVES][64];
    __shared__ RowopMasks64 tree_gates[Q64_TREE_WAVES][QG_WAVE];
    __shared__ uint32_t mask_part[Q64_TREE_THREADS + 2 + Q64_TREE_WAVES];
    // the row-operation table comes into LDS while the mask's counts (or the list's length) are in flight: the draws then index LDS instead of paying a
    // third dependent trip to memory (kernels_qm.hip qm_init_block does the same; `opaque_zero` keeps the uniform loads vector loads, i.e. not waited
    // for where they are issued)
    __shared__ uint32_t tree_table[QG_TREE_TABLE_MAX];
    const bool table_fits = a.num_actions <= QG_TREE_TABLE_MAX;
    uint32_t opaque_zero;
    asm("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    // the finished envs: the bits of the mask the step before left (if it did), then the entries of the list (compacted from raised flags; or, after a fused
    // launch, the few envs that were reset inside it and final again after their first step)
    DoneMaskShare share;
    uint32_t hint_v = 0;
    const uint32_t lcount_v = a.list_count[opaque_zero];
    if (a.mask) {  // every workgroup counts the mask (a hint word with another number: nobody finished)
        hint_v = done_mask_hint(a.mask, a.B)[opaque_zero];
        done_mask_load<Q64_TREE_THREADS>(a.mask, a.B, a.mask_words, share);
    }
    asm volatile("" ::: "memory");
    if (table_fits)
        for (uint32_t i = threadIdx.x; i < a.num_actions; i += blockDim.x) tree_table[i] = a.rowops[i];
    const bool mask_empty = !a.mask || (uint32_t)__builtin_amdgcn_readfirstlane((int)hint_v) != a.mask_epoch;
    const uint32_t mcount = mask_empty ? 0u : done_mask_scan<Q64_TREE_THREADS>(share, mask_part);
    const uint32_t len = mcount + (uint32_t)__builtin_amdgcn_readfirstlane((int)lcount_v);
    if (a.count_out && blockIdx.x == 0 && threadIdx.x == 0) *a.count_out = len;  // (host memory: sizes the next launches' tree grid, qgym_api.cpp reset_tree_grid)
    if (threadIdx.x == 0) phase_stamp(a.kclk, a.kclk_waves, 0);  // the count is known
    if (!tree_takes(len, a.n_draws)) return;
    // (no tickets with InitArgs::zero_count: workgroup 0 zeroes the idle list's length; the barrier inside -- workgroups with work -- also makes the table visible)
    const uint32_t count = list_count_take(a.list_count, len, (uint64_t)(len < a.tree_grid ? len : a.tree_grid) * Q64_TREE_THREADS, blockIdx.x, a.zero_count);
    const uint32_t N = a.N;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    // entry blockIdx.x of the list, then + tree_grid, ...: the launch has that many tree workgroups for a list of any (tree) length
    for (uint32_t item = blockIdx.x; item < count; item += a.tree_grid) {
        if (item != blockIdx.x) __syncthreads();  // (the previous round's LDS has been read)
        // (the first entry from the thread that holds it in its share of the mask: a barrier, no search; the barrier also makes the table visible)
        const uint64_t env = item >= mcount ? a.list[item - mcount]
                           : item == blockIdx.x ? done_mask_find<Q64_TREE_THREADS>(a.mask, a.mask_words, share, mask_part, item)
                                                : done_mask_nth<Q64_TREE_THREADS>(a.mask, a.mask_words, mask_part, item);
        if (threadIdx.x == 0) phase_stamp(a.kclk, a.kclk_waves, 1);  // the env is known
        if (threadIdx.x < QG_WAVE) after.ahead(env);  // (wave 0 finishes the env)
        uint64_t myrow = 0;
        // q64_init_finish with the wave's 64 lanes: lane s holds the row of slot s (scramble_tree64 runs on the transpose), stores its 8 bytes of the
        // env's tile and compares with the identity's; lane 0 writes the scalars (reset_internals, clifford.rs:272-283)
        if (!scramble_tree64<NS, Q64_TREE_WAVES>(a, env, myrow, prod, tree_gates, table_fits ? tree_table : a.rowops, [N](uint32_t k) -> uint64_t { return q64_identity_word<NS, HAS_Z>((int)k, N); })) continue;
        phase_stamp(a.kclk, a.kclk_waves, 2);  // the scramble is done (wave 0)
        float step_penalty = 0.0f;
        if constexpr (After::WAVE_STEP) {  // the env's first step (q64_step1_with, on lanes: lane s holds the row of slot s; the gate's qubits are the same on every lane)
            const int64_t act = after.act;
            if (act >= 0 && act < (int64_t)after.a.num_actions) {
                const GateEntry g = after.g;
                step_penalty = g.penalty;
                const uint32_t q0 = g.ops & 63u, q1 = (g.ops >> 6) & 63u, m = (g.ops >> 12) & 0xFFFFu;
                if (m != Q64_IDENTITY) {
                    const uint32_t s0 = HAS_Z ? 2u * q0 : q0, s1 = HAS_Z ? 2u * q1 : q1;  // the slots of x[q0], x[q1] (z: the next one)
                    auto row_of = [&](uint32_t slot) -> uint64_t {
                        const int l = __builtin_amdgcn_readfirstlane((int)slot);
                        return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)myrow, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(myrow >> 32), l) << 32);
                    };
                    const uint64_t x0 = row_of(s0), x1 = row_of(s1), z0 = HAS_Z ? row_of(s0 + 1u) : 0ull, z1 = HAS_Z ? row_of(s1 + 1u) : 0ull;
                    auto mix = [&](uint32_t k) -> uint64_t {
                        const uint32_t b = m >> (4 * k);
                        uint64_t o = ((0ull - (uint64_t)(b & 1u)) & x0) ^ ((0ull - (uint64_t)((b >> 2) & 1u)) & x1);
                        if (HAS_Z) o ^= ((0ull - (uint64_t)((b >> 1) & 1u)) & z0) ^ ((0ull - (uint64_t)((b >> 3) & 1u)) & z1);
                        return o;
                    };
                    const uint64_t nx0 = mix(0), nx1 = mix(2), nz0 = HAS_Z ? mix(1) : 0ull, nz1 = HAS_Z ? mix(3) : 0ull;
                    // q1's rows first, then q0's (q0's value wins when q0 == q1, as in q64_apply)
                    myrow = lane == s1 ? nx1 : myrow;
                    if (HAS_Z) myrow = lane == s1 + 1u ? nz1 : myrow;
                    myrow = lane == s0 ? nx0 : myrow;
                    if (HAS_Z) myrow = lane == s0 + 1u ? nz0 : myrow;
                }
            }
        }
        const uint32_t le = (uint32_t)(env & (QG_WAVE - 1));
        uint64_t *tile = reinterpret_cast<uint64_t *>(reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Q64Rows<NS>::G * 64));
        if (lane < (uint32_t)NS) tile[((uint64_t)(lane >> 1) * 64 + le) * 2 + (lane & 1u)] = myrow;  // group lane / 2, the env's 16-byte piece, its low or high word
        const uint64_t differs = (uint64_t)__ballot(lane < (uint32_t)NS && myrow != q64_identity_word<NS, HAS_Z>((int)lane, N));
        if (lane != 0) continue;
        uint64_t bad = differs;
        if constexpr (HAS_Z) {  // bit j: slot 2j or 2j + 1 differs
            uint64_t t = (differs | (differs >> 1)) & 0x5555555555555555ull;
            t = (t | (t >> 1)) & 0x3333333333333333ull;
            t = (t | (t >> 2)) & 0x0F0F0F0F0F0F0F0Full;
            t = (t | (t >> 4)) & 0x00FF00FF00FF00FFull;
            t = (t | (t >> 8)) & 0x0000FFFF0000FFFFull;
            bad = (t | (t >> 16)) & 0x00000000FFFFFFFFull;
        }
        const bool solved = differs == 0;
        if (a.bad) reinterpret_cast<uint64_t *>(a.bad)[env] = bad;
        if constexpr (After::WAVE_STEP) {  // reset_internals (clifford.rs:272-283), then the step's bookkeeping (clifford.rs:342-346) on top of it: the values the two calls leave
            const StepArgs &sa = after.a;
            const int32_t depth = a.depth_value > 0 ? a.depth_value - 1 : 0;
            const float reward = (solved ? 1.0f : 0.0f) - step_penalty;
            const bool fin = depth == 0 || solved;
            if (sa.rewards_seq) sa.rewards_seq[env] = reward;
            if (sa.dones_seq) sa.dones_seq[env] = (uint8_t)fin;
            a.depth[env] = depth;
            a.success[env] = (uint8_t)solved;
            a.reward[env] = reward;
            a.done[env] = (uint8_t)fin;
            if (fin) {  // (rare: one atomic per env that is final again after its first step)
                const uint32_t slot = atomicAdd(sa.done_count, 1u);
                if (slot < sa.B) sa.done_list[slot] = (uint32_t)env;
            }
        } else {
        a.depth[env] = a.depth_value;
        a.success[env] = (uint8_t)solved;
        a.reward[env] = solved ? 1.0f : 0.0f;
        a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
        }
        a.inverted[env] = (uint8_t)((HAS_Z && a.check_symplectic) ? Q64_FLAG_SYMPLECTIC : 0u);  // identity + gates: symplectic (q64_init_finish, mode 2)
        uint32_t wave_fault = 0, wave_sol_n = 0;
        if constexpr (After::WAVE_STEP) {  // clifford.rs:334-340: the step's entry in the fresh episode's solution log
            if (after.a.flags & F_TRACK) {
                if (after.a.sol_cap) sol_at(after.a, env, wave_sol_n++) = sol_word_framed(after.act, false);
                else wave_fault |= 8u;
            }
        }
        a.error[env] = wave_fault;
        a.sol_len[env * 2] = (int32_t)wave_sol_n;
        a.sol_len[env * 2 + 1] = 0;
        if (a.layers) {
            const LayerRec lay = layer_rec(a.layers, env, a.layers_len);
            for (uint32_t i = 0; i + 2 < a.layers_len; ++i) lay[i] = -1;
            lay[a.layers_len - 2] = 0;
            lay[a.layers_len - 1] = 0;
        }
        phase_stamp(a.kclk, a.kclk_waves, 3);  // everything is stored
        if constexpr (!After::WAVE_STEP) after(env);
    }
}

// qg_vec_reset_done in ONE launch: the first InitArgs::tree_grid workgroups are trees, the workgroups behind them (one per 64 envs) apply the same test to the same
// count and have nothing to do when the trees took the list; when the list is too long or the scramble too short the trees leave it alone and those workgroups
// take it.  (Two launches until round 5: q64_reset_tree_kernel, then q64_init_kernel -- 1.3 us of launch boundary and a kernel with nothing to do.)
template <int NS, bool HAS_Z>
__global__ __launch_bounds__(Q64_TREE_THREADS) void q64_reset_done_kernel(InitArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    if (blockIdx.x < a.tree_grid) q64_reset_tree_body<NS, HAS_Z>(a);
    else (void)q64_init_body<NS, HAS_Z, Q64_TREE_THREADS, true>(a, blockIdx.x - a.tree_grid);
}
// the first step of an env a reset lane / wave of q64_reset_step_kernel has just written (q64_init_body / q64_reset_tree_body: `After`)
// (WAVE: the tree's wave takes the step on the rows in its lanes -- not with layer weights, whose metric reads the record the reset has just written)
template <int NS, bool HAS_Z, bool FEAT, bool WAVE = !FEAT>
struct Q64FirstStep {
    static constexpr bool WAVE_STEP = WAVE;
    const StepArgs &a;
    int64_t act;
    GateEntry g;
    __device__ void ahead(uint64_t env) {  // (nothing of this depends on the reset)
        act = load_action(a.actions, env, a.flags & F_ACT64);
        g = GateEntry{0u, 0.0f};
        if (act >= 0 && act < (int64_t)a.num_actions) g = a.gates[act];
    }
    __device__ void operator()(uint64_t env) const {
        // the fresh episode's stores -- in the tree, by the wave's other lanes too -- are in the L2 before this lane's loads of them are issued, and those
        // loads do not take a line this CU read earlier (a step workgroup's neighbours of this env).  Not __threadfence(): its release half writes the whole
        // L2's dirty lines back (the L2s of the XCDs are not coherent with each other) -- 26 us a launch instead of 11
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (q64_step1_with<NS, HAS_Z, FEAT>(a, env, act, g)) {  // (rare: one atomic per env that is final again after its first step)
            const uint32_t slot = atomicAdd(a.done_count, 1u);
            if (slot < a.B) a.done_list[slot] = (uint32_t)env;
        }
    }
};
// qg_vec_reset_done followed by qg_vec_step in ONE launch (qg_vec_reset_done_step), as kernels_qm.hip qm_reset_step_kernel: the grid is [tree workgroups][step
// workgroups][the per-lane / 16-lane reset workgroups].  Which envs are being reset is read from the mask of is_final bits the PREVIOUS step left (plus the
// rare entries of the list) and nobody writes during this launch; this launch writes the OTHER mask.  A reset env's first step is taken by the lane that has just
// written its fresh episode (q64_step1_body, reading it back: its own stores, made visible by a fence), and if the env is final again after it, that lane
// appends it to the OTHER list.  Results are those of the two calls.
struct Q64ResetStepArgs {
    InitArgs reset;
    StepArgs step;
    uint32_t step_blocks;
};
template <int NS, bool HAS_Z, bool FEAT, bool WAVE = !FEAT>
__global__ __launch_bounds__(Q64_TREE_THREADS) void q64_reset_step_kernel(Q64ResetStepArgs ra) {
    KernelClock kclk(ra.step.kclk, ra.step.kclk_waves);  // device_common.hpp
    const StepArgs &a = ra.step;
    const uint32_t b = blockIdx.x, trees = ra.reset.tree_grid;
    if (b >= trees && b < trees + ra.step_blocks) {  // a step workgroup
        const uint64_t env = (uint64_t)(b - trees) * blockDim.x + threadIdx.x;
        uint64_t resets = env < a.B ? ra.reset.mask[env >> 6] : 0ull;  // (one word per wave)
        uint32_t relisted = ra.reset.list_count[0];
        relisted = relisted < a.B ? relisted : (uint32_t)a.B;
        for (uint32_t i = 0; i < relisted; ++i) {  // (wave-uniform; rare: envs reset in the previous launch and final again after their first step)
            const uint32_t e = ra.reset.list[i];
            if ((e >> 6) == (uint32_t)(env >> 6)) resets |= 1ull << (e & 63u);
        }
        bool fin = false;
        if (env < a.B && !((resets >> (env & 63u)) & 1ull)) fin = q64_step1_body<NS, HAS_Z, FEAT>(a, env);
        done_mask_store(a.done_mask, a.B, fin, env, a.done_epoch);  // (an env being reset: bit clear)
        return;
    }
    Q64FirstStep<NS, HAS_Z, FEAT, WAVE> first_step{a, 0, GateEntry{0u, 0.0f}};
    if (b < trees) q64_reset_tree_body<NS, HAS_Z>(ra.reset, first_step);
    else (void)q64_init_body<NS, HAS_Z, Q64_TREE_THREADS, true>(ra.reset, b - trees - ra.step_blocks, first_step);
}
template <int NS, bool HAS_Z>
static hipError_t q64_launch_reset_step(const Q64ResetStepArgs &ra, hipStream_t s) {
    const unsigned grid = ra.reset.tree_grid + ra.step_blocks + grid_for(ra.reset.B, 64);
    if (ra.step.flags & F_LAYERS) hipLaunchKernelGGL((q64_reset_step_kernel<NS, HAS_Z, true, false>), dim3(grid), dim3(Q64_TREE_THREADS), 0, s, ra);
    else if (ra.step.flags & F_TRACK) hipLaunchKernelGGL((q64_reset_step_kernel<NS, HAS_Z, true, true>), dim3(grid), dim3(Q64_TREE_THREADS), 0, s, ra);  // (solution log: the wave writes the entry)
    else hipLaunchKernelGGL((q64_reset_step_kernel<NS, HAS_Z, false>), dim3(grid), dim3(Q64_TREE_THREADS), 0, s, ra);
    return hipGetLastError();
}
template <int NS, bool HAS_Z>
static hipError_t q64_launch_init(const InitArgs &a, hipStream_t s) {
    const unsigned all = grid_for(a.B, 64);
    // a list (or mask) of finished envs whose scrambles are long enough for trees: one launch, the tree workgroups first.  Its lists are read without tickets
    // (InitArgs::zero_count: qgym_api.cpp rotates the handle's lists)
    if (a.mode == 2 && a.list && a.coop && a.n_draws >= 64u && a.tree_grid && a.zero_count)
        hipLaunchKernelGGL((q64_reset_done_kernel<NS, HAS_Z>), dim3((unsigned)a.tree_grid + all), dim3(Q64_TREE_THREADS), 0, s, a);
    else if (a.mode == 2)
        hipLaunchKernelGGL((q64_init_kernel<NS, HAS_Z, true>), dim3(all), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((q64_init_kernel<NS, HAS_Z>), dim3(all), dim3(64), 0, s, a);
    return hipGetLastError();
}

// ns = slots: CliffordEnv 2 * (N rounded up to 4); LinearFunctionEnv N rounded up to 8
#define Q64_DISPATCH(FN, ARGS)                        \
    if (has_z) {                                      \
        switch (ns) {                                 \
        case 40: return FN<40, true>(ARGS, s);        \
        case 48: return FN<48, true>(ARGS, s);        \
        case 56: return FN<56, true>(ARGS, s);        \
        case 64: return FN<64, true>(ARGS, s);        \
        }                                             \
    } else {                                          \
        switch (ns) {                                 \
        case 40: return FN<40, false>(ARGS, s);       \
        case 48: return FN<48, false>(ARGS, s);       \
        case 56: return FN<56, false>(ARGS, s);       \
        case 64: return FN<64, false>(ARGS, s);       \
        }                                             \
    }                                                 \
    return hipErrorInvalidValue;

hipError_t q64_step(const StepArgs &a, uint32_t ns, bool has_z, hipStream_t s) {
    if (!a.B) return hipSuccess;
    Q64_DISPATCH(q64_launch_step, a)
}
hipError_t q64_reset_step(const InitArgs &reset, const StepArgs &step, uint32_t ns, bool has_z, hipStream_t s) {
    if (!reset.B) return hipSuccess;
    Q64ResetStepArgs ra;
    ra.reset = reset;
    ra.step = step;
    ra.step_blocks = grid_for(step.B, Q64_TREE_THREADS);
    Q64_DISPATCH(q64_launch_reset_step, ra)
}
hipError_t q64_init(const InitArgs &a, uint32_t ns, bool has_z, hipStream_t s) {
    if (!a.B) return hipSuccess;
    Q64_DISPATCH(q64_launch_init, a)
}
hipError_t q64_export(const ObsArgs &a, uint32_t ns, bool has_z, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(q64_export_kernel, dim3(grid_for(a.B * a.D, 256)), dim3(256), 0, s, a, ns, has_z ? 1u : 0u);
    return hipGetLastError();
}

}  // namespace qg
