// kernels_qm.hip -- TILE layout ("qubit machine"): thread-per-env kernels for GF(2) matrices of at
// most 32 rows: CliffordEnv N <= 16 (the headline shape) and LinearFunctionEnv 8 < N <= 32.
//
// Reference semantics: same as kernels_rows.hip --
//   Clifford::step rust/src/envs/clifford.rs:321-347, gates :89-133, solved :136-145;
//   LinearFunction::step rust/src/envs/linear_function.rs:302-328, cx/swap :62-83.
//
// Why a second layout: the ROWS layout spends ~19 wave-instructions per env-step (8 lanes per env)
// and is instruction-issue-bound on MI355X.  Here one lane owns one whole env, so every wave
// instruction advances 64 envs and nothing crosses lanes: no shuffles, no ballots.  Three kernels:
//   qm_step1_kernel   one env.step() per launch without add_inverts (the hot path): holds nothing, gathers and
//                     scatters the gate's <= 2 row groups, `solved` from an incremental mask (~1.5 wave
//                     instructions per env-step);
//   qm_step_kernel    fused rollouts and add_inverts: the env's <= 32 row words live in VGPRs (~4.5);
//   qm_init_kernel    set_state / reset / reset_done (scramble on LDS-resident rows).
//
// Memory (TILE layout): envs are grouped in tiles of 64 (one wavefront).  A tile stores its rows as
// R/4 "row groups" of 1 KiB: group g holds, for lane l, the uint4 {slot 4g .. 4g+3} of env
// tile*64 + l.  Every load/store instruction of a wave is therefore one contiguous, fully
// coalesced 1 KiB access (16 B per lane), and a tile is one contiguous R*256-byte block.
// Slots: LinearFunctionEnv row j is slot j.  CliffordEnv interleaves the two rows of a qubit: X-type
// row j is slot 2j, Z-type row N+j is slot 2j+1, so a 16-byte group holds everything two
// neighbouring qubits own and a one-qubit gate dirties ONE group (measured: the store phase, not
// the loads, is what a step pays for; one dirty group instead of two is -16 % per launch at
// B = 65 536 and -29 % at B = 2^20).  NXP = N rounded up to a multiple of 4; unused slots hold zero.
//
// An action is (q0, q1, M): the four rows {X[q0], Z[q0], X[q1], Z[q1]} are replaced by GF(2)
// combinations of themselves given by the 4x4 bit matrix M (H, S, SX, CX, CZ, SWAP and "no gate"
// are all of this form).  Per-lane row selection is a compare + conditional-move sweep over the
// register file, the identity test an xor/or sweep.
#include <cstdlib>

#include "device_common.hpp"
#include "qgym_plan.hpp"
#include "qm_step1.hpp"

namespace qg {

// ops word of the TILE layout: [0:5) q0, [5:10) q1, [10:26) M (QM_IDENTITY: qm_step1.hpp).
// M bit 4*k + i: output k takes input i, inputs/outputs ordered {X[q0], Z[q0], X[q1], Z[q1]}.

template <int NXP, bool HAS_Z>
struct QmRows {
    static constexpr int R = HAS_Z ? 2 * NXP : NXP;  // row slots per env
    static constexpr int G = R / 4;                  // 16-byte groups per env
    static constexpr int xs(int j) { return HAS_Z ? 2 * j : j; }      // slot of X-type row j
    static constexpr int zs(int j) { return 2 * j + 1; }              // slot of Z-type row N+j
    uint32_t r[R];
};

template <int NXP, bool HAS_Z>
__device__ inline void qm_load(const uint4 *tile, uint32_t lane, QmRows<NXP, HAS_Z> &s) {
#pragma unroll
    for (int g = 0; g < QmRows<NXP, HAS_Z>::G; ++g) {
        const uint4 q = tile[g * 64 + lane];
        s.r[4 * g + 0] = q.x; s.r[4 * g + 1] = q.y; s.r[4 * g + 2] = q.z; s.r[4 * g + 3] = q.w;
    }
}

template <int NXP, bool HAS_Z>
__device__ inline void qm_identity(QmRows<NXP, HAS_Z> &s, uint32_t N) {
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        s.r[QmRows<NXP, HAS_Z>::xs(j)] = (uint32_t)j < N ? 1u << j : 0u;
        if (HAS_Z) s.r[QmRows<NXP, HAS_Z>::zs(j)] = (uint32_t)j < N ? (1u << N) << j : 0u;
    }
}

// CFState::solved / LFState::solved (clifford.rs:136-145, linear_function.rs:91-100)
template <int NXP, bool HAS_Z>
__device__ inline bool qm_solved(const QmRows<NXP, HAS_Z> &s, uint32_t N) {
    uint32_t acc[4] = {0, 0, 0, 0};  // four independent or-chains (ILP: one wave per SIMD at B = 65536)
    const uint32_t zb = 1u << N;
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        acc[j & 1] |= s.r[QmRows<NXP, HAS_Z>::xs(j)] ^ ((uint32_t)j < N ? 1u << j : 0u);
        if (HAS_Z) acc[2 + (j & 1)] |= s.r[QmRows<NXP, HAS_Z>::zs(j)] ^ ((uint32_t)j < N ? zb << j : 0u);
    }
    return ((acc[0] | acc[1]) | (acc[2] | acc[3])) == 0;
}

// bit j: qubit j's rows (CliffordEnv) / row j (LinearFunctionEnv) differ from the identity's
template <int NXP, bool HAS_Z>
__device__ inline uint32_t qm_badmask(const QmRows<NXP, HAS_Z> &s, uint32_t N) {
    uint32_t bad = 0;
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const uint32_t ix = (uint32_t)j < N ? 1u << j : 0u, iz = (uint32_t)j < N ? (1u << N) << j : 0u;
        bool differs = s.r[QmRows<NXP, HAS_Z>::xs(j)] != ix;
        if (HAS_Z) differs = differs || s.r[QmRows<NXP, HAS_Z>::zs(j)] != iz;
        bad |= (uint32_t)differs << j;
    }
    return bad;
}

// r[q] for a per-lane q: tree_select (qm_step1.hpp), a binary select tree on the bits of q (depth log2(n) instead of an n-long
// dependent chain).

// apply one action; returns a bit mask of the 16-byte groups that were written
template <int NXP, bool HAS_Z>
__device__ inline uint32_t qm_apply(QmRows<NXP, HAS_Z> &s, uint32_t ops) {
    const uint32_t q0 = ops & 31u, q1 = (ops >> 5) & 31u, m = (ops >> 10) & 0xFFFFu;
    uint32_t xs[NXP], zs[NXP];
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        xs[j] = s.r[QmRows<NXP, HAS_Z>::xs(j)];
        zs[j] = HAS_Z ? s.r[QmRows<NXP, HAS_Z>::zs(j)] : 0u;
    }
    const uint32_t x0 = tree_select<NXP>(xs, q0), x1 = tree_select<NXP>(xs, q1);
    const uint32_t z0 = HAS_Z ? tree_select<NXP>(zs, q0) : 0u, z1 = HAS_Z ? tree_select<NXP>(zs, q1) : 0u;
    // GF(2) mix: out_k = xor_i M[k][i] * in_i   (-(bit) is an all-ones / all-zeros lane mask)
    auto mix = [&](uint32_t k) -> uint32_t {
        const uint32_t b = m >> (4 * k);
        uint32_t o = (0u - (b & 1u)) & x0;
        o ^= (0u - ((b >> 2) & 1u)) & x1;
        if (HAS_Z) {
            o ^= (0u - ((b >> 1) & 1u)) & z0;
            o ^= (0u - ((b >> 3) & 1u)) & z1;
        }
        return o;
    };
    const uint32_t nx0 = mix(0), nx1 = mix(2);
    const uint32_t nz0 = HAS_Z ? mix(1) : 0u, nz1 = HAS_Z ? mix(3) : 0u;
    const uint32_t w0 = q0, w1 = q1;
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const bool h0 = w0 == (uint32_t)j, h1 = w1 == (uint32_t)j;
        // two flat selects (q0's value wins when q0 == q1): keeps this a v_cndmask sweep
        uint32_t vx = s.r[QmRows<NXP, HAS_Z>::xs(j)];
        vx = h1 ? nx1 : vx;
        vx = h0 ? nx0 : vx;
        s.r[QmRows<NXP, HAS_Z>::xs(j)] = vx;
        if (HAS_Z) {
            uint32_t vz = s.r[QmRows<NXP, HAS_Z>::zs(j)];
            vz = h1 ? nz1 : vz;
            vz = h0 ? nz0 : vz;
            s.r[QmRows<NXP, HAS_Z>::zs(j)] = vz;
        }
    }
    // groups of 4 slots: two qubits (CliffordEnv, X/Z interleaved) or four rows (LinearFunctionEnv)
    const uint32_t gsh = HAS_Z ? 1u : 2u;
    const uint32_t dirty = (1u << (q0 >> gsh)) | (1u << (q1 >> gsh));
    return m == QM_IDENTITY ? 0u : dirty;  // "no gate" writes nothing back
}


// ------------------------------------------------------------------------------------------
// add_inverts (clifford.rs:262-270): state := state^-1 with probability 1/2 after every step.
//
// The reference runs Gauss-Jordan (clifford.rs:147-170).  The inverse of a matrix is unique, so
// any correct algorithm is bit-exact.  CliffordEnv states are symplectic whenever they come from
// reset() or from set_state(get_state(clifford)) -- then M^-1 = Omega M^T Omega with
// Omega = [[0 I],[I 0]]: a 32x32 bit transpose in registers (~5x16 butterfly steps) instead of
// O(D^3) row operations.  Whether a state is symplectic is established once, when it is installed
// (qm_init_kernel verifies M * (Omega M^T Omega) == I and records it in bit 1 of the env's
// `inverted` byte; gates and inversions preserve the property).  Envs holding any other matrix
// take the Gauss-Jordan path below, which also detects singular matrices (the reference panics).
//
// Both work in "slot space": a square R x R matrix (R = 2*NXP) whose row i and column i belong to
// the same slot; logical Z-columns N..2N-1 are moved to bit positions NXP..NXP+N-1 and back.

__device__ inline uint32_t qm_cols_to_slots(uint32_t w, uint32_t N, uint32_t nxp) {
    const uint32_t xm = (1u << N) - 1u;
    return (w & xm) | (((w >> N) & xm) << nxp);
}
__device__ inline uint32_t qm_cols_from_slots(uint32_t w, uint32_t N, uint32_t nxp) {
    const uint32_t xm = (1u << N) - 1u;
    return (w & xm) | (((w >> nxp) & xm) << N);
}

// in-register 32x32 bit-matrix transpose (a[r] bit c <-> a[c] bit r): five butterfly stages; the two coarse ones exchange whole
// bytes, which v_perm_b32 does in one instruction per output word instead of the three of the shift / mask / xor form
__device__ inline void qm_transpose32(uint32_t (&a)[32]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {  // stage 16: high halfword of a[k] <-> low halfword of a[k + 16]
        const uint32_t lo = a[k], hi = a[k + 16];
        a[k] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        a[k + 16] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {  // stage 8: odd bytes of a[k] <-> even bytes of a[k + 8]
        if ((k & 8) == 0) {
            const uint32_t lo = a[k], hi = a[k + 8];
            a[k] = __builtin_amdgcn_perm(hi, lo, 0x06020400u);
            a[k + 8] = __builtin_amdgcn_perm(hi, lo, 0x07030501u);
        }
    }
#pragma unroll
    for (int st = 2; st < 5; ++st) {
        const int j = 16 >> st;
        const uint32_t m = st == 2 ? 0x0F0F0F0Fu : st == 3 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if ((k & j) == 0) {
                const uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
                a[k + j] ^= t;
                a[k] ^= t << j;
            }
        }
    }
}

// candidate inverse Omega M^T Omega of a CliffordEnv state, slot space in and out
template <int NXP>
__device__ inline void qm_symplectic_candidate(const uint32_t (&m)[32], uint32_t (&c)[32]) {
    constexpr int R = 2 * NXP;
    uint32_t t[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i] = m[i];
    qm_transpose32(t);
    const uint32_t rmask = R == 32 ? 0xFFFFFFFFu : ((1u << R) - 1u);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < R) {
            const uint32_t w = t[(i + NXP) % R];
            c[i] = ((w >> NXP) | (w << NXP)) & rmask;  // swap the X and Z column halves
        } else {
            c[i] = 0;
        }
    }
}

template <int NXP>
__device__ inline void qm_to_slot_space(const QmRows<NXP, true> &s, uint32_t N, uint32_t (&m)[32]) {
    if (N == (uint32_t)NXP) {  // no padding slots (e.g. the 16-qubit flagship): columns already are slot positions
#pragma unroll
        for (int i = 0; i < 32; ++i)
            m[i] = i < NXP ? s.r[QmRows<NXP, true>::xs(i)] : (i < 2 * NXP ? s.r[QmRows<NXP, true>::zs(i - NXP)] : 0u);
        return;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i)  // slot-space row i: X-type rows first, then Z-type rows
        m[i] = i < NXP ? qm_cols_to_slots(s.r[QmRows<NXP, true>::xs(i)], N, NXP)
                       : (i < 2 * NXP ? qm_cols_to_slots(s.r[QmRows<NXP, true>::zs(i - NXP)], N, NXP) : 0u);
}
template <int NXP>
__device__ inline void qm_from_slot_space(QmRows<NXP, true> &s, uint32_t N, const uint32_t (&m)[32]) {
    if (N == (uint32_t)NXP) {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            s.r[QmRows<NXP, true>::xs(i)] = m[i];
            s.r[QmRows<NXP, true>::zs(i)] = m[NXP + i];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        s.r[QmRows<NXP, true>::xs(i)] = qm_cols_from_slots(m[i], N, NXP);
        s.r[QmRows<NXP, true>::zs(i)] = qm_cols_from_slots(m[NXP + i], N, NXP);
    }
}

// is `c` the inverse of `m`?  (both slot space; unused slots must be all-zero rows of the product)
template <int NXP>
__device__ inline bool qm_is_inverse(const uint32_t (&m)[32], const uint32_t (&c)[32], uint32_t N) {
    constexpr int R = 2 * NXP;
    uint32_t bad = 0;
#pragma unroll 1
    for (int i = 0; i < R; ++i) {
        uint32_t row = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) row = (k == i) ? m[k] : row;  // m[i], i uniform
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) acc ^= (0u - ((row >> j) & 1u)) & c[j];
        const bool real = (uint32_t)(i % NXP) < N;
        bad |= acc ^ (real ? 1u << i : 0u);
    }
    return bad == 0;
}

// Gauss-Jordan in slot space for the envs that are not known to be symplectic.  Row operations
// only (the first row below with the pivot bit is xor-ed into the pivot row instead of swapped
// with it: a different elimination path to the same, unique, inverse).  Returns false for a
// singular matrix -- the reference's `expect("CFState is singular; cannot invert")`.
template <int NXP, bool HAS_Z>
__device__ __noinline__ bool qm_gauss_jordan(QmRows<NXP, HAS_Z> &s, uint32_t N) {
    constexpr int R = QmRows<NXP, HAS_Z>::R;
    uint32_t m[32], v[32];
    if constexpr (HAS_Z) {
        qm_to_slot_space<NXP>(s, N, m);
    } else {  // LinearFunctionEnv: the slots already are a square matrix (linear_function.rs:124-146)
#pragma unroll
        for (int i = 0; i < 32; ++i) m[i] = i < R ? s.r[i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const bool pad = i < R && (uint32_t)(i % NXP) >= N;
        if (pad) m[i] = 1u << i;  // unused slots: a decoupled identity block, removed again below
        v[i] = i < R ? 1u << i : 0u;
    }
    uint32_t singular = 0;
#pragma unroll 1
    for (int col = 0; col < R; ++col) {
        uint32_t pm = 0, pv = 0;  // the pivot row (row `col`; col is wave-uniform)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            pm = (k == col) ? m[k] : pm;
            pv = (k == col) ? v[k] : pv;
        }
        uint32_t need = ((pm >> col) & 1u) - 1u;  // all ones while the diagonal bit is missing
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const uint32_t t = (r > col && r < R) ? (need & (0u - ((m[r] >> col) & 1u))) : 0u;
            pm ^= m[r] & t;
            pv ^= v[r] & t;
            need &= ~t;
        }
        singular |= need;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const uint32_t t = (r != col && r < R) ? (0u - ((m[r] >> col) & 1u)) : 0u;
            m[r] = (r == col) ? pm : (m[r] ^ (pm & t));
            v[r] = (r == col) ? pv : (v[r] ^ (pv & t));
        }
    }
    if (singular) return false;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const bool pad = i < R && (uint32_t)(i % NXP) >= N;
        if (pad) v[i] = 0;
    }
    if constexpr (HAS_Z) {
        qm_from_slot_space<NXP>(s, N, v);
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) s.r[i] = v[i];
    }
    return true;
}

template <int NXP>
__device__ inline void qm_symplectic_inverse(QmRows<NXP, true> &s, uint32_t N) {
    uint32_t m[32], c[32];
    qm_to_slot_space<NXP>(s, N, m);
    qm_symplectic_candidate<NXP>(m, c);
    qm_from_slot_space<NXP>(s, N, c);
}

template <int NXP>
__device__ inline bool qm_check_symplectic(const QmRows<NXP, true> &s, uint32_t N) {
    uint32_t m[32], c[32];
    qm_to_slot_space<NXP>(s, N, m);
    qm_symplectic_candidate<NXP>(m, c);
    return qm_is_inverse<NXP>(m, c, N);
}

// FEAT: compile in the rarely used per-step extras (solution log, layer-weighted metrics); the
// plain instantiation keeps the hot path free of their code and registers.
// SEQ: several steps per launch (fused rollout) and/or per-step reward/done outputs.
// INV: add_inverts (CliffordEnv only in this layout).
// GJ:  with INV, also compile the general Gauss-Jordan inversion (LinearFunctionEnv always; CliffordEnv
//      only when some env holds a non-symplectic matrix -- its mere presence costs the symplectic fast
//      path 3.5 us per launch in register pressure and scratch, so the host picks the variant).
template <int NXP, bool HAS_Z, bool FEAT, bool SEQ, bool INV = false, bool GJ = false>
__global__ __launch_bounds__(256) void qm_step_kernel(StepArgs a) {
    using Rows = QmRows<NXP, HAS_Z>;
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    const bool valid = env < a.B;
    const bool act64 = a.flags & F_ACT64;
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64);

    if (!valid) return;  // no cross-lane operation below: idle lanes of the last tile just leave
    // the action first (the gate-table read depends on it), then the rows and the depth; the
    // table (<= a few KiB) is read straight from global memory: it stays L1/L2 resident, and
    // measured faster than staging it in LDS behind a barrier
    int64_t act = load_action(a.actions, env, act64);
    Rows s;
    qm_load<NXP, HAS_Z>(tile, lane, s);
    int32_t depth = a.depth[env];
    uint32_t iflags = INV ? a.inverted[env] : 0u;

    uint32_t dirty = 0;
    bool solved = false;
    float reward = 0.0f;
    uint32_t fault = 0;
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    int32_t sol_b = (FEAT && INV && (a.flags & F_TRACK)) ? a.sol_len[env * 2 + 1] : 0;

    const uint32_t T = SEQ ? a.T : 1u;
    for (uint32_t t = 0; t < T; ++t) {
        if (SEQ && t) act = load_action(a.actions, (uint64_t)t * a.B + env, act64);
        const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
        GateEntry g = {QM_IDENTITY << 10, 0.0f};
        if (in_range) g = a.gates[act];
        float penalty = g.penalty;
        if (FEAT && (a.flags & F_LAYERS) && in_range)
            penalty = layers_penalty(layer_rec(a.layers, env, 2 * a.N + 2), a.N, a.descs[act], a.w);

        dirty |= qm_apply<NXP, HAS_Z>(s, g.ops);  // apply_gate_to_state (clifford.rs:331)

        if (FEAT && (a.flags & F_TRACK)) {  // clifford.rs:334-340: entries in push order, bit 31 = pushed to solution_inv
            if ((uint32_t)(sol_n + sol_b) < a.sol_cap) {
                const bool inv_frame = INV && (iflags & QM_FLAG_INVERTED);
                sol_at(a, env, (uint32_t)(sol_n + sol_b)) = sol_word_framed(act, inv_frame);
                if (inv_frame) ++sol_b;
                else ++sol_n;
            } else {
                fault |= 8u;
            }
        }
        depth = depth > 0 ? depth - 1 : 0;          // clifford.rs:342
        if constexpr (INV) {                        // maybe_random_invert (clifford.rs:262-270, linear_function.rs:227-235)
            const uint32_t coin = a.coins ? a.coins[(uint64_t)t * a.B + env]
                                          : (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a) + t) >> 63);
            if (coin & 1u) {
                bool fast = false;
                if constexpr (HAS_Z) {
                    if (iflags & QM_FLAG_SYMPLECTIC) {
                        qm_symplectic_inverse<NXP>(s, a.N);
                        fast = true;
                    }
                }
                if (fast) {
                    iflags ^= QM_FLAG_INVERTED;
                    dirty = 0xFFFFFFFFu;
                } else if constexpr (GJ) {
                    if (qm_gauss_jordan<NXP, HAS_Z>(s, a.N)) {
                        iflags ^= QM_FLAG_INVERTED;
                        dirty = 0xFFFFFFFFu;
                    } else {
                        fault |= QG_FAULT_SINGULAR;
                    }
                } else {
                    fault |= QG_FAULT_BAD_STATE;  // unreachable: the host launches the GJ variant whenever such an env may exist
                }
            }
        }
        solved = qm_solved<NXP, HAS_Z>(s, a.N);     // clifford.rs:344
        const float achieved = solved ? 1.0f : 0.0f;
        reward = achieved - penalty;                // clifford.rs:345-346
        if (SEQ && a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
        if (SEQ && a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
    }

#pragma unroll
    for (int g = 0; g < Rows::G; ++g)
        if ((dirty >> g) & 1u) tile[g * 64 + lane] = make_uint4(s.r[4 * g], s.r[4 * g + 1], s.r[4 * g + 2], s.r[4 * g + 3]);
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);  // is_final (clifford.rs:353)
    a.success[env] = (uint8_t)solved;
    if (FEAT && (a.flags & F_TRACK)) {
        a.sol_len[env * 2] = sol_n;
        if (INV) a.sol_len[env * 2 + 1] = sol_b;
    }
    if (INV) a.inverted[env] = (uint8_t)iflags;
    if ((FEAT || INV) && fault) atomicOr(&a.error[env], fault);
    if (a.bad) a.bad[env] = qm_badmask<NXP, HAS_Z>(s, a.N);  // the one-step kernel may run next
}

// ------------------------------------------------------------------------------------------
// The reference-default env.step() (add_inverts = true, clifford.rs:262-270,334-340): TWO lanes per env.
//
// Why: this step is instruction-bound, not memory-bound (apply the gate, and on half of the lanes -- the coin is per env --
// a 32x32 bit transpose).  At 65 536 envs a thread-per-env kernel is ONE wave per SIMD, and one wave alone issues a
// vector instruction every 4 cycles where two waves issue one every 2 (MI355X_MICROARCH.md, cycle constants): splitting an env
// over a lane pair halves the instructions per lane AND doubles the issue rate.  Lane h of the pair owns qubits 8h .. 8h+7: the
// rows {X[q], Z[q]} = groups 4h .. 4h+3 of the tile, 16 row words.
//
// The inverse of a symplectic M is Omega M^T Omega (see above).  Index rows and columns by v = 16 t + q (t = 0: X, 1: Z):
// inv[v][v'] = M[v' ^ 16][v ^ 16], i.e. inv[v] = rot16(T[v ^ 16]) with T the plain 32x32 transpose.  The transpose is five
// butterfly stages, one per index bit, in any order: bit 3 of v is the lane (stage 8 = one DPP exchange with the partner lane +
// one v_perm_b32 per word), bits 0-2 and bit 4 pair words of the same lane; stage 16 is fused with the rot16 and the v ^ 16
// renaming into one v_perm_b32 per word.  ~150 instructions per lane instead of ~600 per env.
// States that are not known to be symplectic (set_state of an arbitrary matrix) keep the thread-per-env Gauss-Jordan variant.
// ------------------------------------------------------------------------------------------
// (LIST: the envs that finish are recorded for the qg_vec_reset_done that follows -- one bit per env, the wave's ballot stored as half a word of
// StepArgs::done_mask; round 4 appended their indices to a list with one atomic per workgroup of 1 024 threads, which cost this kernel 4.2 us of
// its 7.5: half of the CUs idle, the others with sixteen waves each)
template <int NXP, bool FEAT, bool LIST = false, bool DENSE = false>
__global__ __launch_bounds__(256) void qm_inv2_kernel(StepArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t dense_rows[DENSE ? 4 : 1][DENSE ? 32 * 33 : 1];  // DENSE: the inverted envs' rows, [env of the wave][row], pitch 33
    uint32_t *dl = DENSE ? dense_rows[threadIdx.x >> 6] : nullptr;
    QG_PREFETCH_STEP_ARGS(a);
    bool whole = false;  // DENSE: this lane's env was inverted, its rows are parked in dl
    if constexpr (LIST || DENSE) {  // every thread reaches the wave-wide ballot / rewrite; whole lane pairs are in or out together
        bool fin = false;
        if ((tid >> 1) < a.B) fin = qm_inv2_body<NXP / 2, FEAT, DENSE>(a, NXP / 2, tid >> 1, (uint32_t)tid & 1u, nullptr, dl, &whole);  // qm_step1.hpp
        if constexpr (DENSE) qm_inv2_dense_flush(a.dense, dl, (tid - (threadIdx.x & 63u)) >> 1, whole);
        if constexpr (LIST) done_mask_store_pairs(a.done_mask, a.B, fin, tid, a.done_epoch);
    } else {
        if ((tid >> 1) >= a.B) return;  // whole lane pairs leave together
        (void)qm_inv2_body<NXP / 2, FEAT, DENSE>(a, NXP / 2, tid >> 1, (uint32_t)tid & 1u, nullptr, dl, &whole);  // qm_step1.hpp
    }
}

// One step per launch without holding the matrix (the env.step() path without add_inverts).  A gate
// touches the rows of <= 2 qubits (CliffordEnv) / <= 2 rows (LinearFunctionEnv), i.e. <= 2 of the env's
// 16-byte groups: they are gathered and scattered at per-lane addresses, and `solved` comes from the
// incrementally kept `bad` mask (bit j: qubit j's rows / row j differ from the identity's).  ~3x fewer
// instructions than the register-resident kernel, which at one wave per SIMD is what a step costs;
// measured 3.77 -> 3.15 us per step at B = 65 536 and 34.9 -> 30.0 us at B = 2^20 (CliffordEnv 16q).
// LIST: also record the envs that finish, one bit each in StepArgs::done_mask (F_DONE_LIST; its own instantiation: the plain kernel's code stays as it is)
// DENSE (qg_vec_track_dense, N == NXP, D % 16 == 0): the rows the gate rewrote also go to the resident dense int8 observation
template <int NXP, bool HAS_Z, bool FEAT, bool LIST = false, bool DENSE = false>
__global__ __launch_bounds__(256) void qm_step1_kernel(StepArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    using Rows = QmRows<NXP, HAS_Z>;
    constexpr int D16 = DENSE ? Rows::R / 16 : 0;
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    QG_PREFETCH_STEP_ARGS(a);  // qgym_internal.hpp
    if constexpr (LIST) {  // every thread reaches the wave's ballot
        bool fin = false;
        if (env < a.B) fin = qm_step1_body<HAS_Z, FEAT, D16>(a, Rows::G, env, load_action(a.actions, env, a.flags & F_ACT64));
        done_mask_store(a.done_mask, a.B, fin, env, a.done_epoch);
    } else {
        if (env >= a.B) return;
        const int64_t act = load_action(a.actions, env, a.flags & F_ACT64);
        (void)qm_step1_body<HAS_Z, FEAT, D16>(a, Rows::G, env, act);  // qm_step1.hpp
    }
}

// Fused rollout on LDS-resident rows (T steps per launch, plain configuration: no add_inverts, no solution log, default
// weights).  The register-resident fused kernel pays ~300 VALU instructions per env-step for select trees over 32 row
// registers; here the env's rows live in LDS as [slot][lane] (one column per lane: any per-lane slot is conflict-free), so a
// gate is four dynamic-index LDS reads, the 4x4 GF(2) mix and four writes, and `solved` is the incremental mask of the
// one-step kernel.  Actions and gate entries of the next four steps are fetched ahead of the dependent LDS chain.
template <int NXP, bool HAS_Z, bool ACT64>
__global__ __launch_bounds__(256) void qm_fused_lds_kernel(StepArgs a) {
    using Rows = QmRows<NXP, HAS_Z>;
    __shared__ uint32_t lds_rows[4][Rows::R][QG_WAVE];
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    if (env >= a.B) return;  // no cross-lane operation and no barrier below
    uint32_t (*rows)[QG_WAVE] = lds_rows[threadIdx.x >> 6];
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64);
    uint32_t bad;
    {
        Rows s;
        qm_load<NXP, HAS_Z>(tile, lane, s);
        bad = qm_badmask<NXP, HAS_Z>(s, a.N);
#pragma unroll
        for (int k = 0; k < Rows::R; ++k) rows[k][lane] = s.r[k];
    }
    int32_t depth = a.depth[env];
    const uint32_t zb = 1u << a.N;
    float reward = 0.0f;
    bool solved = bad == 0;
    // two-stage prefetch, four steps per stage: while batch k is applied, the gate entries of batch k + 1 are in flight (their
    // actions arrived during batch k - 1) and so are the actions of batch k + 2 -- nothing on the LDS chain waits for global memory
    // every load is unconditional (clamped index, result selected afterwards): behind a branch the compiler waits for each load
    // before issuing the next one, which serialises eight memory round trips per four steps
    auto load_act = [&](uint32_t t) -> int64_t {
        const uint64_t i = (uint64_t)(t < a.T ? t : a.T - 1u) * a.B + env;
        const int64_t v = ACT64 ? reinterpret_cast<const int64_t *>(a.actions)[i] : (int64_t) reinterpret_cast<const int32_t *>(a.actions)[i];
        return t < a.T ? v : -1;
    };
    auto load_gate = [&](int64_t act, GateEntry &g) {
        const bool ok = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
        const GateEntry e = a.gates[ok ? act : 0];
        g.ops = ok ? e.ops : (QM_IDENTITY << 10);
        g.penalty = ok ? e.penalty : 0.0f;
    };
    auto step = [&](uint32_t t, const GateEntry &g) {
        if (t >= a.T) return;
        const uint32_t q0 = g.ops & 31u, q1 = (g.ops >> 5) & 31u, m = (g.ops >> 10) & 0xFFFFu;
        if (m != QM_IDENTITY) {
            const uint32_t sx0 = HAS_Z ? 2u * q0 : q0, sx1 = HAS_Z ? 2u * q1 : q1;
            const uint32_t x0 = rows[sx0][lane], x1 = rows[sx1][lane];
            const uint32_t z0 = HAS_Z ? rows[sx0 + 1u][lane] : 0u, z1 = HAS_Z ? rows[sx1 + 1u][lane] : 0u;
            auto mix = [&](uint32_t k) -> uint32_t {  // out_k = xor_i M[k][i] * in_i
                const uint32_t b = m >> (4 * k);
                uint32_t o = ((0u - (b & 1u)) & x0) ^ ((0u - ((b >> 2) & 1u)) & x1);
                if (HAS_Z) o ^= ((0u - ((b >> 1) & 1u)) & z0) ^ ((0u - ((b >> 3) & 1u)) & z1);
                return o;
            };
            const uint32_t nx0 = mix(0), nx1 = mix(2), nz0 = HAS_Z ? mix(1) : 0u, nz1 = HAS_Z ? mix(3) : 0u;
            // q1's rows first, then q0's (q0's value wins when q0 == q1, as in qm_apply)
            rows[sx1][lane] = nx1;
            if (HAS_Z) rows[sx1 + 1u][lane] = nz1;
            rows[sx0][lane] = nx0;
            if (HAS_Z) rows[sx0 + 1u][lane] = nz0;
            const uint32_t b1 = (uint32_t)(nx1 != (1u << q1) || (HAS_Z && nz1 != (zb << q1)));
            const uint32_t b0 = (uint32_t)(nx0 != (1u << q0) || (HAS_Z && nz0 != (zb << q0)));
            bad = (bad & ~(1u << q1)) | (b1 << q1);
            bad = (bad & ~(1u << q0)) | (b0 << q0);
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
    for (int g = 0; g < Rows::G; ++g)
        tile[g * 64 + lane] = make_uint4(rows[4 * g][lane], rows[4 * g + 1][lane], rows[4 * g + 2][lane], rows[4 * g + 3][lane]);
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);  // is_final (clifford.rs:353)
    a.success[env] = (uint8_t)solved;
    if (a.bad) a.bad[env] = bad;
}

// slot of matrix row `row`
__device__ inline uint32_t qm_slot(uint32_t row, uint32_t N, uint32_t nxp, bool has_z) {
    (void)nxp;
    return has_z ? (row < N ? 2 * row : 2 * (row - N) + 1) : row;
}

// the tail of set_state / reset for one env: rows to the tile, reset_internals (clifford.rs:272-283)
// (RESET_ONLY: the instantiation inside the one-launch reset + step kernels, whose mode is always 2 -- set_state's symplectic check, 32 x 32 transposes in
// registers, stays out of them: the kernels' registers decide how many workgroups share a CU)
template <int NXP, bool HAS_Z, bool RESET_ONLY = false>
__device__ inline void qm_init_finish(const InitArgs &a, uint64_t env, const QmRows<NXP, HAS_Z> &s) {
    using Rows = QmRows<NXP, HAS_Z>;
    const uint32_t lane = (uint32_t)(env & (QG_WAVE - 1));
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64);
    const bool solved = qm_solved<NXP, HAS_Z>(s, a.N);
#pragma unroll
    for (int g = 0; g < Rows::G; ++g) tile[g * 64 + lane] = make_uint4(s.r[4 * g], s.r[4 * g + 1], s.r[4 * g + 2], s.r[4 * g + 3]);
    if (a.bad) a.bad[env] = qm_badmask<NXP, HAS_Z>(s, a.N);
    if constexpr (Rows::R % 16 == 0) {  // qg_vec_track_dense, list resets (a few envs): the env's whole dense observation, row by row
        if (a.dense) {
#pragma unroll
            for (int k = 0; k < Rows::R; ++k) {
                const uint32_t row = HAS_Z ? ((k & 1) ? a.N + (k >> 1) : (k >> 1)) : (uint32_t)k;
                dense_row_store<Rows::R / 16>(a.dense, env, row, s.r[k]);
            }
        }
    }
    a.depth[env] = a.depth_value;  // reset_internals (clifford.rs:272-283)
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
    // bit 1: the state is symplectic (identity and everything reached from it by gates is; an
    // arbitrary set_state matrix is checked once, here), so inversions may use the transpose form
    uint32_t symp = 0;
    if constexpr (HAS_Z) {
        if (a.check_symplectic) {
            bool is_symp = true;  // identity + gates
            if constexpr (!RESET_ONLY) is_symp = a.mode != 1 || qm_check_symplectic<NXP>(s, a.N);
            symp = is_symp ? QM_FLAG_SYMPLECTIC : 0u;
            if (!symp && a.nonsymp_flag) atomicOr(a.nonsymp_flag, 1u);
        }
    }
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

// qm_init_finish for a reset env whose rows sit one per lane (scramble_tree: lane s holds the row of slot s): called by the 64 lanes of a wave.
// Lane s stores its 4 bytes of the tile (four lanes = one 16-byte group) and its row of the dense observation, the `bad` mask and `solved` come
// from one ballot, lane 0 writes the scalars (reset_internals, clifford.rs:272-283; identity + gates is symplectic).
template <int NXP, bool HAS_Z>
__device__ inline void qm_init_finish_wave(const InitArgs &a, uint64_t env, uint32_t myrow) {
    using Rows = QmRows<NXP, HAS_Z>;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1), le = (uint32_t)(env & (QG_WAVE - 1)), N = a.N;
    uint32_t *tile = reinterpret_cast<uint32_t *>(reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64));
    const bool slot = lane < (uint32_t)Rows::R;
    const uint32_t j = HAS_Z ? lane >> 1 : lane;  // the slot's qubit / row
    const uint32_t ident = (slot && j < N) ? ((HAS_Z && (lane & 1u)) ? (1u << N) << j : 1u << j) : 0u;
    if (slot) tile[((lane >> 2) * 64u + le) * 4u + (lane & 3u)] = myrow;
    const uint64_t differs = __ballot(slot && myrow != ident);
    if constexpr (Rows::R % 16 == 0) {  // qg_vec_track_dense: the env's whole dense observation, a row per lane
        if (a.dense && slot) dense_row_store<Rows::R / 16>(a.dense, env, HAS_Z ? ((lane & 1u) ? N + j : j) : lane, myrow);
    }
    if (lane != 0) return;
    uint32_t bad = (uint32_t)differs;
    if constexpr (HAS_Z) {  // bit j: slot 2j or 2j + 1 differs
        uint32_t t = (bad | (bad >> 1)) & 0x55555555u;
        t = (t | (t >> 1)) & 0x33333333u;
        t = (t | (t >> 2)) & 0x0F0F0F0Fu;
        t = (t | (t >> 4)) & 0x00FF00FFu;
        bad = (t | (t >> 8)) & 0x0000FFFFu;
    }
    const bool solved = differs == 0;
    if (a.bad) a.bad[env] = bad;
    a.depth[env] = a.depth_value;
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
    a.inverted[env] = (uint8_t)((HAS_Z && a.check_symplectic) ? QM_FLAG_SYMPLECTIC : 0u);
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

// qm_init_finish_wave followed by the env's first step (qg_vec_reset_done_step, plain configuration: no solution log, no layer metrics), on
// the wave that holds the fresh rows one per lane: the gate's four rows are read across the lanes (the action is uniform), mixed, and put
// back before anything is stored, so the step costs no trip to memory for what the reset has just written -- as two calls the lane that
// finished the reset loaded depth, mask, action, gate entry and row groups back, three dependent trips (1.4 us of the launch's 11).
// `act`, `g`: the env's action and its gate entry (requested during the scramble).  Results are those of qm_init_finish_wave + qm_step1_body.
// Returns is_final (on every lane).
// INV (qm_reset_inv2_step_kernel): the reference-default step -- the solution log's entry, then maybe_random_invert on `coin` -- i.e. qm_inv2_body.
template <int NXP, bool HAS_Z, bool INV = false>
__device__ inline bool qm_init_finish_wave_step(const InitArgs &a, const StepArgs &sa, uint64_t env, uint32_t myrow, int64_t act, GateEntry g, uint32_t coin = 0) {
    using Rows = QmRows<NXP, HAS_Z>;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1), le = (uint32_t)(env & (QG_WAVE - 1)), N = a.N;
    const bool in_range = act >= 0 && act < (int64_t)sa.num_actions;  // gateset.get(action) (clifford.rs:324)
    const float penalty = in_range ? g.penalty : 0.0f;
    const uint32_t q0 = g.ops & 31u, q1 = (g.ops >> 5) & 31u, m = (g.ops >> 10) & 0xFFFFu;
    if (in_range && m != QM_IDENTITY) {  // (uniform) qm_step1_body's gate on rows held across the lanes
        const uint32_t s0 = HAS_Z ? 2u * q0 : q0, s1 = HAS_Z ? 2u * q1 : q1;
        const uint32_t x0 = (uint32_t)__shfl((int)myrow, (int)s0), x1 = (uint32_t)__shfl((int)myrow, (int)s1);
        const uint32_t z0 = HAS_Z ? (uint32_t)__shfl((int)myrow, (int)(s0 + 1u)) : 0u, z1 = HAS_Z ? (uint32_t)__shfl((int)myrow, (int)(s1 + 1u)) : 0u;
        auto mix = [&](uint32_t k) -> uint32_t {  // out_k = xor_i M[k][i] * in_i
            const uint32_t b = m >> (4 * k);
            uint32_t o = ((0u - (b & 1u)) & x0) ^ ((0u - ((b >> 2) & 1u)) & x1);
            if (HAS_Z) o ^= ((0u - ((b >> 1) & 1u)) & z0) ^ ((0u - ((b >> 3) & 1u)) & z1);
            return o;
        };
        const uint32_t nx0 = mix(0), nx1 = mix(2), nz0 = HAS_Z ? mix(1) : 0u, nz1 = HAS_Z ? mix(3) : 0u;
        // q1's rows first, then q0's (q0's value wins when q0 == q1, as in qm_apply)
        myrow = lane == s1 ? nx1 : myrow;
        if (HAS_Z) myrow = lane == s1 + 1u ? nz1 : myrow;
        myrow = lane == s0 ? nx0 : myrow;
        if (HAS_Z) myrow = lane == s0 + 1u ? nz0 : myrow;
    }
    uint32_t iflags = (HAS_Z && a.check_symplectic) ? QM_FLAG_SYMPLECTIC : 0u;
    if constexpr (INV && HAS_Z) {  // maybe_random_invert (clifford.rs:262-270) on rows held across the lanes: M^-1 = Omega M^T Omega, i.e. with sigma(k) = k +- N the
        if (coin & 1u) {           // new row i is {bit sigma(i) of row sigma(j)}_j: the ballot of bit sigma(i) over the lanes, its odd slots (Z rows) then its even ones (X rows)
            const uint32_t q = lane >> 1, t = lane & 1u;
            const uint32_t my_p = (lane < (uint32_t)Rows::R && q < N) ? (t ? q : N + q) : 0xFFu;  // sigma of this lane's logical row t N + q
            uint32_t mine = 0;
#pragma unroll
            for (uint32_t p = 0; p < 32u; ++p) {
                const uint32_t b = (uint32_t)__ballot((myrow >> p) & 1u);  // (lanes past the slots hold zero rows)
                mine = my_p == p ? b : mine;
            }
            auto even_bits = [](uint32_t x) -> uint32_t {  // bit 2k -> bit k
                x &= 0x55555555u;
                x = (x | (x >> 1)) & 0x33333333u;
                x = (x | (x >> 2)) & 0x0F0F0F0Fu;
                x = (x | (x >> 4)) & 0x00FF00FFu;
                return (x | (x >> 8)) & 0x0000FFFFu;
            };
            myrow = even_bits(mine >> 1) | (even_bits(mine) << N);
            iflags ^= QM_FLAG_INVERTED;
        }
    }
    uint32_t *tile = reinterpret_cast<uint32_t *>(reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64));
    const bool slot = lane < (uint32_t)Rows::R;
    const uint32_t j = HAS_Z ? lane >> 1 : lane;  // the slot's qubit / row
    const uint32_t ident = (slot && j < N) ? ((HAS_Z && (lane & 1u)) ? (1u << N) << j : 1u << j) : 0u;
    if (slot) tile[((lane >> 2) * 64u + le) * 4u + (lane & 3u)] = myrow;
    const uint64_t differs = __ballot(slot && myrow != ident);
    if constexpr (Rows::R % 16 == 0) {  // qg_vec_track_dense: the env's whole dense observation, a row per lane
        if (a.dense && slot) dense_row_store<Rows::R / 16>(a.dense, env, HAS_Z ? ((lane & 1u) ? N + j : j) : lane, myrow);
    }
    const bool solved = differs == 0;                                   // clifford.rs:344
    const int32_t depth = a.depth_value > 0 ? a.depth_value - 1 : 0;    // clifford.rs:317, 342
    const bool fin = depth == 0 || solved;                              // clifford.rs:353
    if (lane != 0) return fin;
    uint32_t bad = (uint32_t)differs;
    if constexpr (HAS_Z) {  // bit j: slot 2j or 2j + 1 differs
        uint32_t t = (bad | (bad >> 1)) & 0x55555555u;
        t = (t | (t >> 1)) & 0x33333333u;
        t = (t | (t >> 2)) & 0x0F0F0F0Fu;
        t = (t | (t >> 4)) & 0x00FF00FFu;
        bad = (t | (t >> 8)) & 0x0000FFFFu;
    }
    const float reward = (solved ? 1.0f : 0.0f) - penalty;  // clifford.rs:345-346
    if (sa.rewards_seq) sa.rewards_seq[env] = reward;
    if (sa.dones_seq) sa.dones_seq[env] = (uint8_t)fin;
    if (a.bad) a.bad[env] = bad;
    a.depth[env] = depth;
    a.success[env] = (uint8_t)solved;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)fin;
    a.inverted[env] = (uint8_t)iflags;
    uint32_t fault = 0, sol_n = 0;
    if (sa.flags & F_TRACK) {  // clifford.rs:334-340: the fresh episode's first entry, pushed in the frame the env is in before the coin (not inverted)
        if (sa.sol_cap) sol_at(sa, env, sol_n++) = sol_word_framed(act, false);
        else fault |= 8u;
    }
    a.error[env] = fault;
    a.sol_len[env * 2] = (int32_t)sol_n;
    a.sol_len[env * 2 + 1] = 0;
    if (a.layers) {
        const LayerRec lay = layer_rec(a.layers, env, a.layers_len);
        for (uint32_t i = 0; i + 2 < a.layers_len; ++i) lay[i] = -1;
        lay[a.layers_len - 2] = 0;
        lay[a.layers_len - 1] = 0;
    }
    return fin;
}

// set_state / reset / reset_done for the thread's env: the work of one workgroup of the init kernel.  `vblock`: the workgroup's index among the
// workgroups doing this work.  Returns true on the threads that finished an env (`env`: which one) -- every thread of the full-batch
// modes, one lane per env of the cooperative list scrambles.
// `after(env, stepped, fin)`: called on every lane that has finished an env (one lane per env on the list paths; a tree workgroup walks the
// list in rounds -- entries vblock, vblock + tree_grid, ... -- so it may be called several times).
// `sa` (qm_reset_step_kernel, plain configuration): the tree path also takes the env's first step (qm_init_finish_wave_step) and says so in
// `stepped`, with is_final in `fin`; the other paths leave the step to `after`.
// PAIR (qm_reset_inv2_step_kernel; list paths only): the env's first step is the reference-default one (add_inverts).  Tree path with `sa`: on the rows the wave holds
// (qm_init_finish_wave_step<.., INV>); otherwise by TWO adjacent lanes reading the fresh episode back (qm_inv2_body): `after(env, h)` is called on an even
// lane (h = 0) and the odd lane next to it (h = 1), both active.
struct NoAfter { __device__ void operator()(uint64_t, bool, bool) const {} };
// LEAN: 0 = everything (set_state too); 1 = RESET_ONLY below (lists: inside the one-launch kernels and qg_vec_reset_done's launch); 2 = mode 2 without a list (qg_vec_reset
// of the whole batch, only_done from the flags): no set_state code either, but the lane-per-env scramble on all four waves as in 0 (69 VGPRs: four workgroups
// per CU by their 32 KB of rows instead of three by 148 VGPRs)
template <int NXP, bool HAS_Z, bool PAIR = false, int LEAN = 0, typename After = NoAfter>
__device__ __forceinline__ void qm_init_block(const InitArgs &a, uint32_t vblock, const StepArgs *sa = nullptr, After after = After()) {
    using Rows = QmRows<NXP, HAS_Z>;
    constexpr bool RESET_ONLY = LEAN == 1, NO_SET_STATE = LEAN != 0;
    // reset scramble (device_common.hpp): the rows live in LDS (wave-private, [slot][lane]: conflict-free for
    // any per-lane slot), so a gate is dynamic-index reads and writes instead of a sweep over 32 registers.
    // RESET_ONLY (inside the one-launch reset + step kernels): only the workgroup's FIRST wave runs the lane-per-env scramble, 64 envs per workgroup (the
    // launcher sizes the grid for it), so the kernel keeps one wave's rows in LDS -- or what the 16-lane scramble of four waves needs, if that is more -- instead
    // of four: 40 KB per workgroup made three workgroups a CU's limit, and the step workgroups queued behind the trees for slots
    constexpr uint32_t FLAT_WAVES = RESET_ONLY ? 1u : 4u;
    constexpr size_t COOP_BYTES = scramble_coop_lds_bytes<uint32_t, Rows::R>(4), FULL_BYTES = sizeof(uint32_t) * 4 * Rows::R * QG_WAVE;
    constexpr bool coop_fits = COOP_BYTES <= FULL_BYTES;  // (the criterion of the four-wave form: qgym_plan.hpp describes one kernel family)
    constexpr size_t LEAN_BYTES = FULL_BYTES / 4 > (coop_fits ? COOP_BYTES : 0) ? FULL_BYTES / 4 : COOP_BYTES;
    __shared__ uint32_t lds_raw[(RESET_ONLY ? (LEAN_BYTES > 1024 ? LEAN_BYTES : 1024) : FULL_BYTES) / sizeof(uint32_t)];  // (>= 8 x 32 words: scramble_tree's products)
    uint32_t(*lds_rows)[Rows::R][QG_WAVE] = reinterpret_cast<uint32_t(*)[Rows::R][QG_WAVE]>(lds_raw);
    const uint64_t tid = (uint64_t)vblock * (FLAT_WAVES * QG_WAVE) + threadIdx.x;  // (lane-per-env path; RESET_ONLY: threads past the first wave leave before it)
    uint64_t env = tid;
    if (LEAN != 2 && a.list) {  // qg_vec_reset_done, compacted: thread i owns the i-th finished env
        static_assert(coop_fits == plan::tile_coop_fits(Rows::R, 4), "qgym_plan.hpp must describe this kernel");
        // scramble_tree's LDS: the gates' masks, and the row-operation table, which comes in while the list length is still in flight (the
        // draws then index LDS instead of paying a third dependent trip to memory)
        __shared__ uint4 tree_gates[LEAN == 2 ? 1 : 4][QG_WAVE];                 // (LEAN 2 is never launched with a list: its LDS is the rows)
        __shared__ uint32_t tree_table[LEAN == 2 ? 1 : QG_TREE_TABLE_MAX];
        const bool table_fits = a.coop && a.num_actions <= QG_TREE_TABLE_MAX;
        // three trips to memory, all in flight together: the list's length, the entry this workgroup would scramble as a tree (any index below B
        // is readable) and the table.  The first two are uniform, and a uniform load is waited for where it is issued (the compiler moves its
        // result to a scalar register at once) -- three round trips one after the other.  An index the compiler cannot see through makes them
        // ordinary vector loads, the clobber keeps them up here, and their values become scalars after the table has been requested.
        uint32_t opaque_zero;
        asm("v_mov_b32 %0, 0" : "=v"(opaque_zero));
        const uint32_t count_v = a.list_count[opaque_zero];
        const uint32_t entry_v = a.coop ? a.list[(vblock < a.B ? vblock : 0u) + opaque_zero] : 0u;
        // ... and, when the step before left its finishers as a mask (InitArgs::mask), this thread's share of the mask's words
        __shared__ uint32_t mask_part[LEAN == 2 ? 1 : 257 + 5];
        DoneMaskShare share;
        uint32_t hint_v = 0;
        if (a.mask) {
            hint_v = done_mask_hint(a.mask, a.B)[opaque_zero];
            done_mask_load(a.mask, a.B, a.mask_words, share);
        }
        asm volatile("" ::: "memory");
        if (table_fits)
            for (uint32_t i = threadIdx.x; i < a.num_actions; i += blockDim.x) tree_table[i] = a.rowops[i];
        // (the hint: no wave of the launch that wrote the mask had a finisher -- nothing to sum, no barriers; workgroup-uniform)
        const bool mask_empty = !a.mask || (uint32_t)__builtin_amdgcn_readfirstlane((int)hint_v) != a.mask_epoch;
        const uint32_t mcount = mask_empty ? 0u : done_mask_scan(share, mask_part);  // (two barriers: the table is visible after them too)
        const uint32_t lcount = (uint32_t)__builtin_amdgcn_readfirstlane((int)count_v);
        const uint32_t count_now = mcount + lcount;
        if (a.count_out && vblock == 0 && threadIdx.x == 0) *a.count_out = count_now;  // (host memory: sizes the next launches' tree grid)
        if (threadIdx.x == 0) phase_stamp(a.kclk, a.kclk_waves, 0);  // the count is known
        // the envs to reset: the mask's set bits in ascending order, then the list's entries
        auto entry = [&](uint32_t i) -> uint32_t { return i < mcount ? done_mask_nth(a.mask, a.mask_words, mask_part, i) : a.list[i - mcount]; };
        // (a block past the list may see the count already zeroed: it has no work either way)
        const plan::ResetPath path = plan::list_reset_path(count_now, a.n_draws, a.B, a.coop != 0, coop_fits);  // qgym_plan.hpp
        const bool tree = path == plan::RP_TREE;
        // this kernel is the list's only reader.  The barrier inside list_count_take (workgroups with work) is also the one that makes the table visible
        const uint64_t with_work = tree ? (uint64_t)(count_now < a.tree_grid ? count_now : a.tree_grid) * QG_TREE_THREADS
                                        : (uint64_t)count_now * (path == plan::RP_COOP ? QG_COOP_LANES : 4u / FLAT_WAVES);  // (threads of 256-thread workgroups with work)
        const uint32_t count = list_count_take(a.list_count, count_now, with_work, vblock, a.zero_count);
        if (tree) {  // few finished envs, long scrambles: a workgroup each, the matrix by columns, the gate sequence cut in eight (scramble_tree)
            const uint32_t N = a.N;
            // entry vblock of the list, then vblock + tree_grid, ...: the launch has tree_grid workgroups for a list of any (tree) length.  (The first
            // round straight-line, the later ones in a loop: as one loop the first round came out 0.3 us slower.)
            auto round = [&](uint64_t e) __attribute__((always_inline)) {
                uint32_t myrow = 0;
                // the first step's action is requested now (a vector load: see above), its gate entry between the chain and the products
                int64_t act = 0;
                GateEntry ge{QM_IDENTITY << 10, 0.0f};
                uint32_t coin = 0;
                if (sa) act = load_action(sa->actions, e + opaque_zero, sa->flags & F_ACT64);
                if constexpr (PAIR) {  // the env's coin (qm_inv2_body): given, or the handle's counter RNG
                    if (sa) coin = sa->coins ? sa->coins[e + opaque_zero] : (uint32_t)(rng_draw(sa->seed ^ 0x636F696Eull, sa->env_base + e, step_clock(*sa)) >> 63);
                }
                const bool finisher = scramble_tree<Rows::R>(a, e, myrow, reinterpret_cast<uint32_t(*)[32]>(&lds_rows[0][0][0]), tree_gates, table_fits ? tree_table : nullptr,
                                                             [N](uint32_t k) -> uint32_t {
                        const uint32_t j = HAS_Z ? k >> 1 : k;
                        return j < N ? ((HAS_Z && (k & 1u)) ? (1u << N) << j : 1u << j) : 0u;
                    }, [&]() {
                        if (sa && act >= 0 && act < (int64_t)sa->num_actions) ge = sa->gates[act];
                    });
                if (!finisher) return;  // the 64 lanes of wave 0 go on
                bool fin = false;
                if (sa) fin = qm_init_finish_wave_step<NXP, HAS_Z, PAIR>(a, *sa, e, myrow, act, ge, coin);
                else qm_init_finish_wave<NXP, HAS_Z>(a, e, myrow);
                if (threadIdx.x == 0) phase_stamp(a.kclk, a.kclk_waves, 4);  // stored
                if constexpr (PAIR) {
                    if (sa) {
                        if ((threadIdx.x & (QG_WAVE - 1)) == 0) after(e, 0u, true, fin);
                    } else if ((threadIdx.x & (QG_WAVE - 1)) < 2u) {
                        after(e, threadIdx.x & 1u, false, false);
                    }
                } else {
                    if ((threadIdx.x & (QG_WAVE - 1)) == 0) after(e, sa != nullptr, fin);
                }
            };
            if (vblock >= count) return;
            // (workgroup-uniform) the first entry: from the list's prefetched word, or from the thread that holds it in its share of the mask
            const uint32_t first = !a.mask ? (uint32_t)__builtin_amdgcn_readfirstlane((int)entry_v)
                                   : vblock < mcount ? done_mask_find(a.mask, a.mask_words, share, mask_part, vblock) : a.list[vblock - mcount];
            round((uint32_t)__builtin_amdgcn_readfirstlane((int)first));
            for (uint32_t item = vblock + a.tree_grid; item < count; item += a.tree_grid) {
                __syncthreads();  // (the previous round's LDS has been read)
                round(entry(item));
            }
            return;
        }
        if (path == plan::RP_COOP) {  // few finished envs: 16 lanes each
            const uint32_t N = a.N;
            const uint32_t *rows = scramble_coop<uint32_t, Rows::R>(a, count, &lds_rows[0][0][0], env, [N](uint32_t k) -> uint32_t {
                const uint32_t j = HAS_Z ? k >> 1 : k;
                return j < N ? ((HAS_Z && (k & 1u)) ? (1u << N) << j : 1u << j) : 0u;
            }, vblock, entry);
            if constexpr (PAIR) {  // lanes 0 and 1 of every 16-lane group with an entry stay: lane 0 finishes the reset, then both take the step
                const uint32_t sl = threadIdx.x & (QG_COOP_LANES - 1);
                if (((uint64_t)vblock * blockDim.x + threadIdx.x) / QG_COOP_LANES >= count || sl >= 2u) return;
                if (rows) {
                    Rows s;
#pragma unroll
                    for (int k = 0; k < Rows::R; ++k) s.r[k] = rows[k];
                    qm_init_finish<NXP, HAS_Z, NO_SET_STATE>(a, env, s);
                }
                after(env, sl, false, false);
                return;
            } else {
                if (!rows) return;
                Rows s;
#pragma unroll
                for (int k = 0; k < Rows::R; ++k) s.r[k] = rows[k];
                qm_init_finish<NXP, HAS_Z, NO_SET_STATE>(a, env, s);
                after(env, false, false);
                return;
            }
        }
        if (RESET_ONLY && threadIdx.x >= QG_WAVE) return;  // (behind every barrier of the list paths: the lane-per-env scramble is the first wave's)
        if constexpr (PAIR) {  // one lane per env for the reset; then the wave's envs one after the other on lanes 0 and 1 (no lane leaves before that)
            const bool has = tid < count;
            if (!__ballot(has)) return;
            const uint32_t mine = has ? entry((uint32_t)tid) : 0u;
            if (has) {
                Rows s;
                qm_identity<NXP, HAS_Z>(s, a.N);
                uint32_t(*rows)[QG_WAVE] = lds_rows[threadIdx.x >> 6];
                const uint32_t L = threadIdx.x & (QG_WAVE - 1);
#pragma unroll
                for (int sl = 0; sl < Rows::R; ++sl) rows[sl][L] = s.r[sl];
                scramble_flat<uint32_t>(rows, L, a, mine);
#pragma unroll
                for (int sl = 0; sl < Rows::R; ++sl) s.r[sl] = rows[sl][L];
                qm_init_finish<NXP, HAS_Z, NO_SET_STATE>(a, mine, s);
            }
            // ... 32 envs at a time: lanes 2 i and 2 i + 1 take the env of lane 32 p + i (one env after the other on lanes 0 and 1 cost a wave 64 steps in a row:
            // 273 us a pair when half of the batch finishes in every step, episodes of two steps)
            const uint64_t owners = __ballot(has);
            const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
#pragma unroll 1
            for (uint32_t p = 0; p < 2u; ++p) {
                const uint32_t src = 32u * p + (lane >> 1);
                const uint32_t e = (uint32_t)__shfl((int)mine, (int)src);
                if ((owners >> src) & 1ull) after((uint64_t)e, lane & 1u, false, false);
            }
            return;
        }
        if (!PAIR && RESET_ONLY && a.flags_current && 4ull * count > a.B && a.n_draws <= 8u) {  // (a curriculum's first difficulties: episodes of a few steps)  // more than a quarter of the batch, in a launch of its own: thread = env, its own flag (no search, no
                                                                              // entry load; a shorter list is better off in full waves: 131 against 89 us at 10 % of 65 536 envs x 256 gates)
            if (env >= a.B || !a.done[env]) return;
        } else {
            if (tid >= count) return;
            env = entry((uint32_t)tid);
        }
    } else {
        if constexpr (PAIR || RESET_ONLY) return;  // (those instantiations are only launched with a list: set_state's code stays out of them)
        if (env >= a.B) return;
        if (a.only_done && !a.done[env]) return;  // live episodes keep running
    }
    Rows s;
    qm_identity<NXP, HAS_Z>(s, a.N);
    if (!NO_SET_STATE && a.mode == 1) {  // set_state (clifford.rs:299-304)
#pragma unroll
        for (int sl = 0; sl < Rows::R; ++sl) {
            const uint32_t j = HAS_Z ? (uint32_t)sl >> 1 : (uint32_t)sl;
            const uint32_t row = (HAS_Z && (sl & 1)) ? a.N + j : j;
            uint32_t w = 0;
            if (j < a.N) {
                if (a.format == QG_FMT_PACKED) {
                    w = reinterpret_cast<const uint32_t *>(a.src)[env * a.src_stride + row];
                    if (a.D < 32) w &= (1u << a.D) - 1u;
                } else if (a.format == QG_FMT_BITS) {  // the entry stream as bits (pack_bitstream): i64 / u8 set_state at streaming rate
                    w = (uint32_t)bits_window(reinterpret_cast<const uint64_t *>(a.src), env * a.src_stride + (uint64_t)row * a.D, a.D);
                } else if (a.format == QG_FMT_I64) {
                    const int64_t *p = reinterpret_cast<const int64_t *>(a.src) + env * a.src_stride + (uint64_t)row * a.D;
                    for (uint32_t c = 0; c < a.D; ++c) w |= (uint32_t)(p[c] > 0) << c;
                } else {
                    const int8_t *p = reinterpret_cast<const int8_t *>(a.src) + env * a.src_stride + (uint64_t)row * a.D;
                    for (uint32_t c = 0; c < a.D; ++c) w |= (uint32_t)(p[c] > 0) << c;
                }
            }
            s.r[sl] = w;
        }
    } else if (a.mode == 2) {  // reset scramble (clifford.rs:306-316)
        uint32_t(*rows)[QG_WAVE] = lds_rows[threadIdx.x >> 6];
        const uint32_t L = threadIdx.x & (QG_WAVE - 1);
#pragma unroll
        for (int sl = 0; sl < Rows::R; ++sl) rows[sl][L] = s.r[sl];
        scramble_flat<uint32_t>(rows, L, a, env);
#pragma unroll
        for (int sl = 0; sl < Rows::R; ++sl) s.r[sl] = rows[sl][L];
    }
    qm_init_finish<NXP, HAS_Z, NO_SET_STATE>(a, env, s);
    if constexpr (!PAIR) after(env, false, false);  // (PAIR: the list paths above have returned; set_state and whole resets are not followed by a step)
}


// RESET_ONLY: qg_vec_reset_done with a list (mode 2): the instantiation without set_state's code, the lane-per-env scramble on the first wave of every workgroup
// (69 instead of 148 VGPRs, 17 instead of 40 KB of LDS: idle workgroups of a tree launch do not hold a third of a CU each)
template <int NXP, bool HAS_Z, int LEAN = 0>
__global__ __launch_bounds__(256) void qm_init_kernel(InitArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    qm_init_block<NXP, HAS_Z, false, LEAN>(a, blockIdx.x);
}

// qg_vec_reset_done followed by qg_vec_step in ONE launch (qg_vec_reset_done_step): the grid's first `reset_blocks` workgroups are the
// reset's (qm_init_block on the list the PREVIOUS step left; they are the launch's long pole -- scramble, then the env's step on the lane
// that finished it -- so they are dispatched first), the others step the envs whose episode goes on.  Nothing is handed over inside the launch: which envs are being reset is read from the mask of
// is_final bits the previous step left and nobody writes during this launch (the user-visible `done` array is written
// by both halves); this launch writes the OTHER mask, and a reset env that is final again after its first step goes to the OTHER list.  Results are
// those of the two calls.
struct ResetStepArgs {
    InitArgs reset;          // reset.mask: the is_final bits the PREVIOUS step left (bit set = this env is being reset in this launch)
    StepArgs step;           // step.done_mask: the bits this launch leaves (the other buffer)
    // the grid: [reset workgroups 0 .. first_reset) [step workgroups] [reset workgroups first_reset .. reset_blocks).  A list's work sits in its first
    // workgroups: those go first (two per CU at 65 536 envs), the step workgroups take the slots beside them, and the reset workgroups that will most
    // likely find no entry come last -- at three workgroups per CU (40 KB of LDS each) whatever is dispatched late waits for a slot
    uint32_t reset_blocks;
    uint32_t step_blocks;
    uint32_t first_reset;
};
// this workgroup's role: true = a step workgroup (`index` among them), false = a reset workgroup (`index` = vblock)
__device__ inline bool reset_step_role(const ResetStepArgs &ra, uint32_t &index) {
    const uint32_t b = blockIdx.x;
    if (b >= ra.first_reset && b < ra.first_reset + ra.step_blocks) {
        index = b - ra.first_reset;
        return true;
    }
    index = b < ra.first_reset ? b : b - ra.step_blocks;
    return false;
}
// (WAVE: the tree's wave takes a reset env's first step -- not with layer weights, whose metric reads the record the reset has just written; a compile-time choice)
template <int NXP, bool HAS_Z, bool FEAT, bool DENSE, bool WAVE = !FEAT>
__global__ __launch_bounds__(256) void qm_reset_step_kernel(ResetStepArgs ra) {
    KernelClock kclk(ra.step.kclk, ra.step.kclk_waves);  // device_common.hpp
    using Rows = QmRows<NXP, HAS_Z>;
    constexpr int D16 = DENSE ? Rows::R / 16 : 0;
    const StepArgs &a = ra.step;
    QG_PREFETCH_STEP_ARGS(a);  // (the reset's lanes reach their step late: its argument lines are requested now, not one miss after the other then)
    uint32_t role_index;
    if (reset_step_role(ra, role_index)) {
        const uint64_t env = (uint64_t)role_index * blockDim.x + threadIdx.x;
        // which envs the reset workgroups are taking: the bits of the previous step's mask, and (rare) the entries of the list -- envs that were reset in
        // the previous launch and final again after their first step, which that launch's reset lanes could only append
        uint64_t resets = env < a.B ? ra.reset.mask[env >> 6] : 0ull;  // (one word per wave)
        uint32_t relisted = ra.reset.list_count[0];
        relisted = relisted < a.B ? relisted : (uint32_t)a.B;
        for (uint32_t i = 0; i < relisted; ++i) {  // (wave-uniform)
            const uint32_t e = ra.reset.list[i];
            if ((e >> 6) == (uint32_t)(env >> 6)) resets |= 1ull << (e & 63u);
        }
        bool fin = false;
        if (env < a.B && !((resets >> (env & 63u)) & 1ull)) {
            // (the dense rows are written by lane pairs: a lane whose neighbour is being reset writes its rows alone)
            const bool alone = D16 != 0 && ((resets >> ((env ^ 1ull) & 63u)) & 1ull);
            fin = qm_step1_body<HAS_Z, FEAT, D16>(a, Rows::G, env, load_action(a.actions, env, a.flags & F_ACT64), alone);
        }
        done_mask_store(a.done_mask, a.B, fin, env, a.done_epoch);  // (an env being reset: bit clear -- if it is final again after its first step the reset's lane appends it to the list)
        return;
    }
    // (plain configuration: the tree's wave takes the env's first step on the rows it holds -- qm_init_finish_wave_step; otherwise the lane
    // that has just written the env's fresh episode -- state, depth, bad mask, log lengths -- takes it, as qm_step1_body)
    qm_init_block<NXP, HAS_Z, false, 1>(ra.reset, role_index, WAVE ? &a : nullptr, [&](uint64_t env, bool stepped, bool fin) {
        if (!stepped) fin = qm_step1_body<HAS_Z, FEAT, D16>(a, Rows::G, env, load_action(a.actions, env, a.flags & F_ACT64), true);
        if (fin) {  // (rare: one atomic per env that is final again after its first step)
            const uint32_t slot = atomicAdd(a.done_count, 1u);
            if (slot < a.B) a.done_list[slot] = (uint32_t)env;
        }
    });
}

// The same with the reference's default options (add_inverts; CliffordEnv N <= 16, every env symplectic): the step workgroups run qm_inv2_body (two lanes
// per env, 512 workgroups at 65 536 envs).  A reset env's first step: the tree's wave takes it on the rows it holds (gate, log entry, the coin's inversion as 32
// ballots: qm_init_finish_wave_step<.., INV>) unless layer weights are on; the 16-lane and per-lane paths hand the fresh episode to two adjacent lanes that read it
// back (qm_inv2_body).  Round 5's first form read back on the tree path too: 15.9 us a pair against 14.3 as two launches (a tree waits ~3 us for its own stores).
// (WAVE: the tree's wave takes the first step -- a compile-time choice: a pointer to the by-value arguments that may be null at run time makes the compiler keep
// all 616 bytes of them in scratch, and a launch with scratch took 46 us)
template <int NXP, bool FEAT, bool WAVE>
__global__ __launch_bounds__(256) void qm_reset_inv2_step_kernel(ResetStepArgs ra) {
    KernelClock kclk(ra.step.kclk, ra.step.kclk_waves);  // device_common.hpp
    const StepArgs &a = ra.step;
    QG_PREFETCH_STEP_ARGS(a);
    uint32_t role_index;
    if (reset_step_role(ra, role_index)) {
        const uint64_t tid = (uint64_t)role_index * blockDim.x + threadIdx.x, env = tid >> 1;
        uint64_t resets = env < a.B ? ra.reset.mask[env >> 6] : 0ull;  // (a wave's 32 envs share a word)
        uint32_t relisted = ra.reset.list_count[0];  // (see qm_reset_step_kernel)
        relisted = relisted < a.B ? relisted : (uint32_t)a.B;
        for (uint32_t i = 0; i < relisted; ++i) {
            const uint32_t e = ra.reset.list[i];
            if ((e >> 6) == (uint32_t)(env >> 6)) resets |= 1ull << (e & 63u);
        }
        bool fin = false;
        if (env < a.B && !((resets >> (env & 63u)) & 1ull)) fin = qm_inv2_body<NXP / 2, FEAT, false>(a, NXP / 2, env, (uint32_t)tid & 1u, nullptr);  // qm_step1.hpp
        done_mask_store_pairs(a.done_mask, a.B, fin, tid, a.done_epoch);
        return;
    }
    const StepArgs *wave_step = WAVE ? &a : nullptr;  // (not with layer weights: that metric reads the record the reset has just written -- the read-back form)
    // `after(env, h, stepped, fin)`: stepped -- the tree's wave has taken the env's step (is_final in `fin`; one lane calls); else lanes h = 0, 1 take it now
    qm_init_block<NXP, true, true, 1>(ra.reset, role_index, wave_step, [&](uint64_t env, uint32_t h, bool stepped, bool fin) {
        if (!stepped) {
            // the fresh episode (written by this wave's lanes) is in the L2 before the pair's loads of it are issued, and those loads do not take a line this CU read
            // earlier (not __threadfence(): its release half writes the XCD's whole L2 back)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            fin = qm_inv2_body<NXP / 2, FEAT, false>(a, NXP / 2, env, h, nullptr);  // qm_step1.hpp
        }
        if (fin && h == 0) {  // (rare: final again after its first step)
            const uint32_t slot = atomicAdd(a.done_count, 1u);
            if (slot < a.B) a.done_list[slot] = (uint32_t)env;
        }
    });
}

// export: one thread per (env, matrix row).  log2L carries NXP/4, flag bit 31 of D's companion
// field is avoided: has_z is passed through `obs_rows != N` (Clifford: D = 2N rows).
__global__ __launch_bounds__(256) void qm_export_kernel(ObsArgs a, uint32_t nxp, uint32_t has_z) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid / a.D;
    const uint32_t row = (uint32_t)(gid % a.D);
    if (env >= a.B) return;
    const uint32_t R = has_z ? 2 * nxp : nxp, slot = qm_slot(row, a.N, nxp, has_z);
    const uint32_t *tile = reinterpret_cast<const uint32_t *>(a.state) + (env >> 6) * (uint64_t)(R * 64);
    const uint32_t w = tile[((slot >> 2) * 64 + (uint32_t)(env & 63)) * 4 + (slot & 3)];
    if (a.format == QG_FMT_PACKED) {
        reinterpret_cast<uint32_t *>(a.out)[env * a.out_stride + row] = w;
    } else if (a.format == QG_FMT_I64) {
        int64_t *o = reinterpret_cast<int64_t *>(a.out) + env * a.out_stride + (uint64_t)row * a.D;
        for (uint32_t c = 0; c < a.D; ++c) o[c] = (int64_t)((w >> c) & 1u);
    } else {
        int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride + (uint64_t)row * a.D;
        for (uint32_t c = 0; c < a.D; ++c) o[c] = (int8_t)((w >> c) & 1u);
    }
}

// Packed observation (QG_FMT_PACKED with no padding between envs: out_stride == D): a wave turns one tile -- 64 envs, slots interleaved
// per lane -- into the 64 * D contiguous row words of those envs.  Both sides are coalesced (1 KiB tile loads, 1 KiB stores when D % 4 == 0);
// the transposition goes through LDS with a row pitch of D + 1 words (D is even for CliffordEnv: conflict-free across the lanes).
__global__ __launch_bounds__(256) void qm_pack_kernel(ObsArgs a, uint32_t nxp, uint32_t has_z) {
    __shared__ uint32_t lds[4][64 * 33];  // D <= 32 in this layout (uint32 rows)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t tile_idx = (uint64_t)blockIdx.x * 4u + wave;
    const uint32_t R = has_z ? 2 * nxp : nxp, G = R / 4, D = a.D, pitch = D + 1u;
    const uint64_t n_tiles = (a.B + 63u) / 64u;
    uint32_t *my = lds[wave];
    if (tile_idx < n_tiles) {
        const uint4 *tile = reinterpret_cast<const uint4 *>(a.state) + tile_idx * (uint64_t)(G * 64);
        uint4 vs[8];  // R <= 32 slots: all groups in flight at once (a loop over a run-time G waits for each load in turn)
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) vs[g] = g < G ? tile[g * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) {
            if (g >= G) break;
            const uint4 v = vs[g];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                const uint32_t slot = 4u * g + c;
                int32_t row;
                if (has_z) row = (slot >> 1) < a.N ? (int32_t)((slot & 1u) ? a.N + (slot >> 1) : (slot >> 1)) : -1;
                else row = slot < a.N ? (int32_t)slot : -1;
                if (row >= 0) my[lane * pitch + (uint32_t)row] = w[c];
            }
        }
    }
    __syncthreads();
    if (tile_idx >= n_tiles) return;
    const uint64_t env0 = tile_idx * 64u;
    const uint32_t envs = a.B - env0 < 64u ? (uint32_t)(a.B - env0) : 64u;
    uint32_t *out = reinterpret_cast<uint32_t *>(a.out) + env0 * D;
    if ((D & 3u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        const uint32_t n4 = envs * D / 4u;
        for (uint32_t i = lane; i < n4; i += 64u) {
            const uint32_t e = (4u * i) / D, r = (4u * i) % D;
            const uint32_t *p = my + e * pitch + r;
            reinterpret_cast<uint4 *>(out)[i] = make_uint4(p[0], p[1], p[2], p[3]);
        }
    } else {
        const uint32_t n = envs * D;
        for (uint32_t i = lane; i < n; i += 64u) out[i] = my[(i / D) * pitch + (i % D)];
    }
}

// Dense int8 observation (adapters.py:50-54) for matrices of D = 16 * D16 rows without padding slots (CliffordEnv N = 8, 16;
// LinearFunctionEnv N = 16, 32): HBM-write bound (D * D bytes per env against D * 4 read).  A wave owns a tile: its 64 envs' rows come
// in with coalesced 1 KiB loads and are turned to [env][row] order in LDS (pitch D + 1 words: conflict-free both ways); then every
// store instruction of the wave writes 1 KiB CONTIGUOUS bytes of the output (lane i owns 16-byte chunk i), the 64 * D * D bytes of
// the tile front to back: 11.5 us at 65 536 envs (rocprofv3) = 5.8 TB/s of written bytes.  (The version before wrote two 16-byte pieces
// 32 bytes apart per lane -- every store instruction spanned 2 KiB with holes: 17.6 us.  A workgroup per tile, 16 envs per wave, with
// or without non-temporal stores, measured 1-1.6 us slower than a wave per tile.)
template <int D16>
__global__ __launch_bounds__(256) void qm_dense_stream_kernel(ObsArgs a, uint32_t has_z) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    constexpr uint32_t D = 16u * D16, G = D / 4u, PITCH = D + 1u, CPE = D * D16;  // CPE: 16-byte chunks per env
    __shared__ uint32_t lds[4][64 * PITCH];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t n_tiles = (a.B + 63u) / 64u;
    const uint64_t tile_idx = (uint64_t)blockIdx.x * 4u + wave;
    if (tile_idx >= n_tiles) return;  // wave-private LDS, no workgroup barrier below
    uint32_t *my = lds[wave];
    const uint4 *tile = reinterpret_cast<const uint4 *>(a.state) + tile_idx * (uint64_t)(G * 64u);
    uint4 vs[G];
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) vs[g] = tile[g * 64u + lane];
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        const uint32_t w[4] = {vs[g].x, vs[g].y, vs[g].z, vs[g].w};
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t slot = 4u * g + c;
            const uint32_t row = has_z ? ((slot & 1u) ? D / 2u + (slot >> 1) : (slot >> 1)) : slot;
            my[lane * PITCH + row] = w[c];
        }
    }
    __builtin_amdgcn_wave_barrier();  // the LDS region is this wave's own: its writes are ordered before its reads (lgkmcnt), no s_barrier
    const uint64_t env0 = tile_idx * 64u;
    const uint32_t envs = a.B - env0 < 64u ? (uint32_t)(a.B - env0) : 64u;
    uint4 *out = reinterpret_cast<uint4 *>(reinterpret_cast<int8_t *>(a.out) + env0 * (uint64_t)(D * D));
    const uint32_t n_chunks = envs * CPE;
#pragma unroll 8
    for (uint32_t i = lane; i < n_chunks; i += 64u) {
        const uint32_t e = i / CPE, r = (i % CPE) / D16, part = i % D16;
        out[i] = expand16_i8(my[e * PITCH + r] >> (16u * part));
    }
}

// The same for every other matrix size of the layout (D <= 32 rows, any N: CliffordEnv 3..15 qubits, LinearFunctionEnv 9..31): the tile's output
// is still one contiguous run of 64 * D * D bytes, a lane still owns an aligned 16-byte chunk of it, but a chunk's bytes now cross rows and
// envs: the lane walks (env, row, column) byte by byte from one division at its start.  (The row-per-thread export kernel wrote a byte per
// store instruction: 0.6 TB/s.)
__global__ __launch_bounds__(256) void qm_dense_stream_any_kernel(ObsArgs a, uint32_t nxp, uint32_t has_z) {
    constexpr uint32_t PITCH = 33u;
    __shared__ uint32_t lds[4][64 * PITCH];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t n_tiles = (a.B + 63u) / 64u;
    const uint64_t tile_idx = (uint64_t)blockIdx.x * 4u + wave;
    if (tile_idx >= n_tiles) return;  // wave-private LDS, no workgroup barrier below
    uint32_t *my = lds[wave];
    const uint32_t N = a.N, D = a.D, DD = D * D, R = has_z ? 2u * nxp : nxp, G = R / 4u;
    const uint4 *tile = reinterpret_cast<const uint4 *>(a.state) + tile_idx * (uint64_t)(G * 64u);
    uint4 vs[8];  // R <= 32 slots: all groups in flight at once
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) vs[g] = g < G ? tile[g * 64u + lane] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        if (g >= G) break;
        const uint32_t w[4] = {vs[g].x, vs[g].y, vs[g].z, vs[g].w};
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t slot = 4u * g + c, j = has_z ? slot >> 1 : slot;
            if (j < N) my[lane * PITCH + ((has_z && (slot & 1u)) ? N + j : j)] = w[c];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint64_t env0 = tile_idx * 64u;
    const uint32_t envs = a.B - env0 < 64u ? (uint32_t)(a.B - env0) : 64u;
    uint8_t *out = reinterpret_cast<uint8_t *>(a.out) + env0 * (uint64_t)DD;
    const uint32_t n_bytes = envs * DD;
    // (env, row, column) of the lane's first byte by division, once; every later chunk is 1 024 bytes further on: wave-uniform increments
    uint32_t f = 16u * lane;
    uint32_t e = f / DD, rem = f - e * DD, r = rem / D, c = rem - r * D;
    const uint32_t adv_e = 1024u / DD, adv_rem = 1024u - adv_e * DD, adv_r = adv_rem / D, adv_c = adv_rem - adv_r * D;
    for (; f < n_bytes; f += 1024u) {
        // the chunk's 16 entries as bits: the rest of row r, then whole rows, then the head of a last one (D >= 3: at most six pieces)
        uint32_t bits = 0, filled = 0, pe = e, pr = r, pc = c;
        while (filled < 16u && pe < envs) {
            const uint32_t take = (16u - filled) < (D - pc) ? (16u - filled) : (D - pc);
            bits |= ((my[pe * PITCH + pr] >> pc) & ((1u << take) - 1u)) << filled;
            filled += take;
            pc = 0;
            if (++pr == D) {
                pr = 0;
                ++pe;
            }
        }
        const uint4 v = expand16_i8(bits);
        if (n_bytes - f >= 16u) {
            *reinterpret_cast<uint4 *>(out + f) = v;
        } else {  // the ragged last tile's last, partial chunk
            const uint32_t o[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t b = 0; b < n_bytes - f; ++b) out[f + b] = (uint8_t)(o[b >> 2] >> (8u * (b & 3u)));
        }
        c += adv_c;
        if (c >= D) {
            c -= D;
            ++r;
        }
        r += adv_r;
        if (r >= D) {
            r -= D;
            ++e;
        }
        e += adv_e;
    }
}

// Dense observation in the policy's dtype straight from the tiles: one thread per 16-byte output
// chunk (D % (16 / ES) == 0), so a wave's store is 1 KiB contiguous; the 4..16 threads that share a
// row read the same resident word (cache hit).
template <int ES>
__global__ __launch_bounds__(256) void qm_dense_typed_kernel(const uint32_t *state, uint64_t B, uint32_t N, uint32_t D, uint32_t nxp,
                                                             uint32_t has_z, uint32_t chunks_per_row, uint4 *out, uint32_t one) {
    constexpr uint32_t EPC = 16 / ES;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t grow = gid / chunks_per_row;  // env * D + row
    const uint64_t env = grow / D;
    if (env >= B) return;
    const uint32_t row = (uint32_t)(grow - env * D), c0 = (uint32_t)(gid - grow * chunks_per_row) * EPC;
    const uint32_t R = has_z ? 2 * nxp : nxp, slot = qm_slot(row, N, nxp, has_z);
    const uint32_t *tile = state + (env >> 6) * (uint64_t)(R * 64);
    const uint32_t w = tile[((slot >> 2) * 64 + (uint32_t)(env & 63)) * 4 + (slot & 3)];
    out[gid] = expand_chunk<ES>((w >> c0) & ((1u << EPC) - 1u), one);
}

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

template <int NXP, bool HAS_Z>
static hipError_t launch_step(const StepArgs &a, hipStream_t s) {
    const dim3 grid(grid_for(a.B, 256)), block(256);  // 64- and 128-thread blocks measured no faster
    const bool feat = a.flags & (F_TRACK | F_LAYERS);
    const bool seq = a.T != 1 || a.rewards_seq || a.dones_seq;
    const bool list = a.flags & F_DONE_LIST;
    switch (plan::tile_step(a.flags, a.T, a.bad != nullptr, a.rewards_seq || a.dones_seq, a.num_actions, HAS_Z, NXP)) {  // qgym_plan.hpp
    case plan::SK_QM_STEP1: {  // the env.step() path
        const dim3 lgrid = grid, lblock = block;
        if constexpr (QmRows<NXP, HAS_Z>::R % 16 == 0) {
            if (a.dense) {  // qg_vec_track_dense (the host passes it for N == NXP only)
                if (feat && list) hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, true, true, true>), lgrid, lblock, 0, s, a);
                else if (feat) hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, true, false, true>), grid, block, 0, s, a);
                else if (list) hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, false, true, true>), lgrid, lblock, 0, s, a);
                else hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, false, false, true>), grid, block, 0, s, a);
                return hipGetLastError();
            }
        }
        if (feat && list) hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, true, true>), lgrid, lblock, 0, s, a);
        else if (feat) hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, true>), grid, block, 0, s, a);
        else if (list) hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, false, true>), lgrid, lblock, 0, s, a);
        else hipLaunchKernelGGL((qm_step1_kernel<NXP, HAS_Z, false>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    case plan::SK_QM_INV2:  // CliffordEnv with add_inverts, every env symplectic, one step per launch: two lanes per env
        if constexpr (HAS_Z && NXP <= 16) {
            const dim3 grid2(grid_for(2 * a.B, 256)), lgrid2 = grid2, lblock = block;
            if constexpr (NXP == 16) {
                if (a.dense) {  // qg_vec_track_dense (N = 16)
                    if (feat && list) hipLaunchKernelGGL((qm_inv2_kernel<NXP, true, true, true>), lgrid2, lblock, 0, s, a);
                    else if (feat) hipLaunchKernelGGL((qm_inv2_kernel<NXP, true, false, true>), grid2, block, 0, s, a);
                    else if (list) hipLaunchKernelGGL((qm_inv2_kernel<NXP, false, true, true>), lgrid2, lblock, 0, s, a);
                    else hipLaunchKernelGGL((qm_inv2_kernel<NXP, false, false, true>), grid2, block, 0, s, a);
                    return hipGetLastError();
                }
            }
            if (feat && list) hipLaunchKernelGGL((qm_inv2_kernel<NXP, true, true>), lgrid2, lblock, 0, s, a);
            else if (feat) hipLaunchKernelGGL((qm_inv2_kernel<NXP, true>), grid2, block, 0, s, a);
            else if (list) hipLaunchKernelGGL((qm_inv2_kernel<NXP, false, true>), lgrid2, lblock, 0, s, a);
            else hipLaunchKernelGGL((qm_inv2_kernel<NXP, false>), grid2, block, 0, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    case plan::SK_QM_STEP_GJ:  // the thread-per-env inversion variants (fused rollouts, states not known to be symplectic) always carry FEAT and SEQ
        if constexpr (HAS_Z && NXP <= 16) {
            hipLaunchKernelGGL((qm_step_kernel<NXP, HAS_Z, true, true, true, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    case plan::SK_QM_STEP_INV:
        if constexpr (HAS_Z && NXP <= 16) {
            hipLaunchKernelGGL((qm_step_kernel<NXP, HAS_Z, true, true, true, false>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    case plan::SK_QM_FUSED_LDS:  // plain fused rollout (an empty gateset has no table to read: the register-resident kernel handles it)
        if (a.flags & F_ACT64) hipLaunchKernelGGL((qm_fused_lds_kernel<NXP, HAS_Z, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((qm_fused_lds_kernel<NXP, HAS_Z, false>), grid, block, 0, s, a);
        return hipGetLastError();
    case plan::SK_QM_STEP:
        if (feat && seq) hipLaunchKernelGGL((qm_step_kernel<NXP, HAS_Z, true, true>), grid, block, 0, s, a);
        else if (feat) hipLaunchKernelGGL((qm_step_kernel<NXP, HAS_Z, true, false>), grid, block, 0, s, a);
        else if (seq) hipLaunchKernelGGL((qm_step_kernel<NXP, HAS_Z, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((qm_step_kernel<NXP, HAS_Z, false, false>), grid, block, 0, s, a);
        return hipGetLastError();
    default: return hipErrorInvalidValue;
    }
}
template <int NXP, bool HAS_Z>
static hipError_t launch_init(const InitArgs &a, hipStream_t s) {
    // reset_done with a short list: scramble_tree walks the list with InitArgs::tree_grid workgroups (the workgroups
    // past the list leave at once); every other path needs at most B threads
    const bool lean = a.mode == 2 && a.list;  // (a workgroup per 64 envs there)
    uint64_t threads = lean ? 4 * a.B : a.B;
    if (a.list && a.coop && a.n_draws >= 64u) {
        const uint64_t tree_blocks = a.tree_grid;
        if (tree_blocks * QG_TREE_THREADS > threads) threads = tree_blocks * QG_TREE_THREADS;
    }
    if (lean) hipLaunchKernelGGL((qm_init_kernel<NXP, HAS_Z, 1>), dim3(grid_for(threads, 256)), dim3(256), 0, s, a);
    else if (a.mode == 2) hipLaunchKernelGGL((qm_init_kernel<NXP, HAS_Z, 2>), dim3(grid_for(threads, 256)), dim3(256), 0, s, a);  // (the whole batch, or by the flags)
    else hipLaunchKernelGGL((qm_init_kernel<NXP, HAS_Z>), dim3(grid_for(threads, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

#define QM_DISPATCH(FN, ARGS)                                         \
    if (has_z) {                                                      \
        switch (nxp) {                                                \
        case 4: return FN<4, true>(ARGS, s);                          \
        case 8: return FN<8, true>(ARGS, s);                          \
        case 12: return FN<12, true>(ARGS, s);                        \
        case 16: return FN<16, true>(ARGS, s);                        \
        }                                                             \
    } else {                                                          \
        switch (nxp) {                                                \
        case 4: return FN<4, false>(ARGS, s);                         \
        case 8: return FN<8, false>(ARGS, s);                         \
        case 12: return FN<12, false>(ARGS, s);                       \
        case 16: return FN<16, false>(ARGS, s);                       \
        case 20: return FN<20, false>(ARGS, s);                       \
        case 24: return FN<24, false>(ARGS, s);                       \
        case 28: return FN<28, false>(ARGS, s);                       \
        case 32: return FN<32, false>(ARGS, s);                       \
        }                                                             \
    }                                                                 \
    return hipErrorInvalidValue;

hipError_t qm_step(const StepArgs &a, uint32_t nxp, bool has_z, hipStream_t s) {
    if (!a.B) return hipSuccess;
    QM_DISPATCH(launch_step, a)
}
hipError_t qm_init(const InitArgs &a, uint32_t nxp, bool has_z, hipStream_t s) {
    if (!a.B) return hipSuccess;
    QM_DISPATCH(launch_init, a)
}
template <int NXP, bool HAS_Z>
static hipError_t launch_reset_step(const ResetStepArgs &ra, hipStream_t s) {
    const InitArgs &ia = ra.reset;
    uint64_t threads = 4 * ia.B;  // the reset's share of the grid: a workgroup per 64 envs (qm_init_block<.., RESET_ONLY>: the lane-per-env scramble is the first wave's)
    uint64_t tree_blocks = 0;
    if (ia.list && ia.coop && ia.n_draws >= plan::TREE_MIN_DRAWS) {
        tree_blocks = ia.tree_grid;
        if (tree_blocks * QG_TREE_THREADS > threads) threads = tree_blocks * QG_TREE_THREADS;
    }
    ResetStepArgs rb = ra;
    rb.reset_blocks = grid_for(threads, 256);
    // the grid: [the tree workgroups][the step workgroups][the other reset workgroups] (no trees: half of the reset workgroups first)
    rb.first_reset = tree_blocks ? (uint32_t)std::min<uint64_t>(tree_blocks, rb.reset_blocks) : rb.reset_blocks / 2;
    const bool feat = ra.step.flags & (F_TRACK | F_LAYERS), layers = ra.step.flags & F_LAYERS;
    if (ra.step.flags & F_INVERTS) {  // the reference-default step: two lanes per env
        if constexpr (HAS_Z && NXP <= 16) {
            rb.step_blocks = grid_for(2 * ra.step.B, 256);
            const dim3 grid2(rb.reset_blocks + rb.step_blocks);
            if (ra.step.flags & F_LAYERS) hipLaunchKernelGGL((qm_reset_inv2_step_kernel<NXP, true, false>), grid2, dim3(256), 0, s, rb);
            else if (feat) hipLaunchKernelGGL((qm_reset_inv2_step_kernel<NXP, true, true>), grid2, dim3(256), 0, s, rb);
            else hipLaunchKernelGGL((qm_reset_inv2_step_kernel<NXP, false, true>), grid2, dim3(256), 0, s, rb);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    rb.step_blocks = grid_for(ra.step.B, 256);
    const dim3 grid(rb.reset_blocks + rb.step_blocks), block(256);
    if constexpr (QmRows<NXP, HAS_Z>::R % 16 == 0) {
        if (ra.step.dense) {
            if (layers) hipLaunchKernelGGL((qm_reset_step_kernel<NXP, HAS_Z, true, true, false>), grid, block, 0, s, rb);
            else if (feat) hipLaunchKernelGGL((qm_reset_step_kernel<NXP, HAS_Z, true, true, true>), grid, block, 0, s, rb);
            else hipLaunchKernelGGL((qm_reset_step_kernel<NXP, HAS_Z, false, true>), grid, block, 0, s, rb);
            return hipGetLastError();
        }
    }
    if (layers) hipLaunchKernelGGL((qm_reset_step_kernel<NXP, HAS_Z, true, false, false>), grid, block, 0, s, rb);
    else if (feat) hipLaunchKernelGGL((qm_reset_step_kernel<NXP, HAS_Z, true, false, true>), grid, block, 0, s, rb);  // (solution log: the wave writes the entry)
    else hipLaunchKernelGGL((qm_reset_step_kernel<NXP, HAS_Z, false, false>), grid, block, 0, s, rb);
    return hipGetLastError();
}
hipError_t qm_reset_step(const InitArgs &reset, const StepArgs &step, uint32_t nxp, bool has_z, hipStream_t s) {
    if (!step.B) return hipSuccess;
    ResetStepArgs ra;
    ra.reset = reset;
    ra.step = step;
    ra.reset_blocks = ra.step_blocks = ra.first_reset = 0;  // (set by the launcher)
    QM_DISPATCH(launch_reset_step, ra)
}

hipError_t qm_export(const ObsArgs &a, uint32_t nxp, bool has_z, hipStream_t s) {
    if (!a.B) return hipSuccess;
    const uint32_t R = has_z ? 2 * nxp : nxp;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(a.out);
    switch (plan::tile_export(a.format, a.D, R, a.out_stride, (addr & 15) == 0, (addr & 3) == 0)) {  // qgym_plan.hpp
    case plan::EK_DENSE_STREAM: {
        const dim3 grid((unsigned)((a.B + 255) / 256)), block(256);  // a wave per tile of 64 envs
        if (a.D == 32) hipLaunchKernelGGL(qm_dense_stream_kernel<2>, grid, block, 0, s, a, has_z ? 1u : 0u);
        else hipLaunchKernelGGL(qm_dense_stream_kernel<1>, grid, block, 0, s, a, has_z ? 1u : 0u);
        return hipGetLastError();
    }
    case plan::EK_DENSE_STREAM_ANY:
        hipLaunchKernelGGL(qm_dense_stream_any_kernel, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, s, a, nxp, has_z ? 1u : 0u);
        return hipGetLastError();
    case plan::EK_PACK:
        hipLaunchKernelGGL(qm_pack_kernel, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, s, a, nxp, has_z ? 1u : 0u);
        return hipGetLastError();
    default: break;
    }
    hipLaunchKernelGGL(qm_export_kernel, dim3(grid_for(a.B * a.D, 256)), dim3(256), 0, s, a, nxp, has_z ? 1u : 0u);
    return hipGetLastError();
}

hipError_t qm_export_typed(const void *state, uint64_t B, uint32_t N, uint32_t D, uint32_t nxp, bool has_z, void *out, uint32_t es,
                           uint32_t one, hipStream_t s) {
    if (!B) return hipSuccess;
    const uint32_t cpr = D / (16 / es);
    if (cpr == 0 || D % (16 / es) != 0) return hipErrorInvalidValue;
    const dim3 grid(grid_for(B * D * cpr, 256)), block(256);
    const uint32_t *st = reinterpret_cast<const uint32_t *>(state);
    uint4 *o = reinterpret_cast<uint4 *>(out);
    if (es == 1) hipLaunchKernelGGL(qm_dense_typed_kernel<1>, grid, block, 0, s, st, B, N, D, nxp, has_z ? 1u : 0u, cpr, o, one);
    else if (es == 2) hipLaunchKernelGGL(qm_dense_typed_kernel<2>, grid, block, 0, s, st, B, N, D, nxp, has_z ? 1u : 0u, cpr, o, one);
    else hipLaunchKernelGGL(qm_dense_typed_kernel<4>, grid, block, 0, s, st, B, N, D, nxp, has_z ? 1u : 0u, cpr, o, one);
    return hipGetLastError();
}

}  // namespace qg
