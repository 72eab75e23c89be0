// kernels_lfd.hip -- LinearFunctionEnv with add_inverts (the reference's default) for 8 < N <= 64: the LFD ("dual") layout.
//
// Reference semantics: rust/src/envs/linear_function.rs -- LFState::cx / swap (:62-83: cx(c, t): row t ^= row c), solved (:91-100),
// inverse (:124-146, Gauss-Jordan; panics on a singular matrix), LinearFunction::step (:302-328: gate, push to solution /
// solution_inv by `inverted`, depth - 1, maybe_random_invert :227-235, success = solved()), set_state (:279-283), reset (:285-300).
//
// Why a layout of its own.  maybe_random_invert replaces the state by its inverse with probability 1/2 after EVERY step.  A general
// GF(2) matrix has no transpose shortcut (CliffordEnv's tableaux do), and a Gauss-Jordan per inversion is O(N^3 / 32) word operations:
// 12 us per step at 12 qubits, 46 at 32 and 221 at 64 (round 1).  Here an env keeps BOTH matrices, A and V = A^-1:
//     step:    A <- G A   is a row operation on A (<= 2 rows);  V <- V G^-1 = V G  is a COLUMN operation on V
//              (cx(c, t): column c ^= column t;  swap(a, b): columns a, b trade places) -- one or two bit operations on every row word;
//     invert:  A and V trade roles: one flag bit flips, nothing moves.
// The inverse is unique, so this is bit-exact with the reference's Gauss-Jordan; only set_state pays one elimination (and records
// "singular", which the reference discovers at the first inversion: the fault is raised then, linear_function.rs:132-134).
//
// Memory (LFD layout): tiles of 64 envs; a tile holds two REGIONS of RG groups of 1 KiB each (group g of a region = rows RPG g ..
// RPG g + RPG - 1 of that matrix for the tile's 64 envs, 16 B per env: four uint32 rows when N <= 32, two uint64 rows otherwise).
// Bit 0 of the env's `inverted` byte -- the reference's own flag -- says which region is the state; the other one is its inverse.
// Per env two row masks (bit r: row r of that region differs from the identity's): the state's is updated incrementally, the
// inverse's is recomputed while its rows stream through the registers anyway.
//
// Mapping: L lanes per env; lane j streams groups j, j + L, ... of the inverse, lane 0 also gathers / scatters the state's <= 2 touched
// rows and owns the scalars.  L = 1 for uint32 rows (N <= 32: <= 8 groups; measured at 65 536 envs, 12 / 16 / 24 / 32 qubits: 5.1 / 5.6 /
// 6.6 / 7.8 us per step with one lane, 6.4 / 6.8 / 7.4 / 8.1 with two, 8.4 / 8.7 / 9.2 / 9.8 with four -- the scalars' and the gate's
// memory instructions are per wave, so more waves per env cost more than the shorter rows loop saves; holding both matrices in
// registers for N <= 16 (no flag -> region chain) was slower too: 6.0 / 7.1 us), L = 4 for uint64 rows (48 / 64 qubits: 18 / 24 us,
// 512 B read + 512 B written per env-step).
#include <type_traits>

#include "device_common.hpp"

namespace qg {

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

#define LFD_FLAG_INVERTED 1u  // = which region holds the state
#define LFD_FLAG_SINGULAR 4u  // set_state installed a singular matrix: the first inversion faults (linear_function.rs:132-134)

template <bool W64>
struct LfdT {
    using W = typename std::conditional<W64, uint64_t, uint32_t>::type;
    static constexpr uint32_t RPG = W64 ? 2u : 4u;  // rows per 16-byte group
    static __device__ inline W get(const uint4 &v, uint32_t k) {
        if constexpr (W64) return k == 0 ? ((uint64_t)v.x | ((uint64_t)v.y << 32)) : ((uint64_t)v.z | ((uint64_t)v.w << 32));
        else return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
    }
    static __device__ inline void put(uint4 &v, uint32_t k, W w) {
        if constexpr (W64) {
            if (k == 0) { v.x = (uint32_t)w; v.y = (uint32_t)(w >> 32); }
            else { v.z = (uint32_t)w; v.w = (uint32_t)(w >> 32); }
        } else {
            v.x = k == 0 ? w : v.x; v.y = k == 1 ? w : v.y; v.z = k == 2 ? w : v.z; v.w = k == 3 ? w : v.w;
        }
    }
    static __device__ inline W ident(uint32_t r, uint32_t N) { return r < N ? (W)1 << r : (W)0; }
};

template <int L>
__device__ inline uint32_t lanes_or(uint32_t v) {  // OR over the L lanes of an env (DPP quad_perm [1,0,3,2], then [2,3,0,1])
    if constexpr (L >= 2) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    if constexpr (L >= 4) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    return v;
}

// ---- one env.step() per launch ---------------------------------------------------------------------------------------------------
template <bool W64, int L, int GPL>
__global__ __launch_bounds__(256) void lfd_step_kernel(StepArgs a, uint32_t RG) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    using T = LfdT<W64>;
    using W = typename T::W;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = tid / (uint32_t)L;
    const uint32_t j = (uint32_t)tid & (uint32_t)(L - 1);
    QG_PREFETCH_STEP_ARGS(a);
    if (env >= a.B) return;  // whole quads leave together
    const uint32_t N = a.N;
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(2u * RG * 64u) + (env & 63u);
    uint32_t iflags = a.inverted[env];
    const uint32_t ra = iflags & LFD_FLAG_INVERTED, rv = ra ^ 1u;  // regions of the state and of its inverse
    // everything that does not depend on the action is requested first: the inverse's groups of this lane, the scalars
    uint4 v[GPL];
#pragma unroll
    for (int k = 0; k < GPL; ++k) {
        const uint32_t g = j + (uint32_t)L * (uint32_t)k;
        v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (g < RG) v[k] = tile[(rv * RG + g) * 64u];
    }
    uint64_t *bad2 = reinterpret_cast<uint64_t *>(a.bad) + env * 2;
    uint64_t bad_a = bad2[ra];
    int32_t depth = a.depth[env];
    uint32_t coin = a.coins ? a.coins[env] : 0u;
    int32_t nf = (a.flags & F_TRACK) ? a.sol_len[env * 2] : 0, nb = (a.flags & F_TRACK) ? a.sol_len[env * 2 + 1] : 0;
    const int64_t act = load_action(a.actions, env, a.flags & F_ACT64);
    const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (linear_function.rs:305)
    const int64_t ai = in_range ? act : 0;
    const uint32_t desc = a.descs[ai];
    float penalty = in_range ? a.gates[ai].penalty : 0.0f;
    if (!a.coins) coin = (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a)) >> 63);
    if ((a.flags & F_LAYERS) && in_range && j == 0) penalty = layers_penalty(layer_rec(a.layers, env, 2 * N + 2), N, desc, a.w);
    const uint32_t kind = desc & 0xFFu, q0 = (desc >> 8) & 0xFFu, q1 = (desc >> 16) & 0xFFu;
    const bool is_cx = in_range && kind == QG_CX && q0 != q1, is_swap = in_range && kind == QG_SWAP && q0 != q1;  // other kinds: ignored (:241)
    uint32_t fault = 0;

    // ---- the state: A <- G A, rows q0 / q1 (lane 0) ----------------------------------------------------------------------------------
    if (j == 0 && (is_cx || is_swap)) {
        const uint32_t g0 = q0 / T::RPG, g1 = q1 / T::RPG;
        uint4 ua = tile[(ra * RG + g0) * 64u], ub = tile[(ra * RG + g1) * 64u];
        if (g0 == g1) ub = ua;
        const W r0 = T::get(ua, q0 % T::RPG), r1 = T::get(ub, q1 % T::RPG)  /* (ub = ua when the rows share a group; a `c ? ua : ub` reference sends both to scratch) */;
        const W n0 = is_swap ? r1 : r0;       // swap: rows trade places (linear_function.rs:72-83)
        const W n1 = is_swap ? r0 : r1 ^ r0;  // cx(q0, q1): row q1 ^= row q0 (:62-70)
        if (g0 == g1) {
            T::put(ua, q0 % T::RPG, n0);
            T::put(ua, q1 % T::RPG, n1);
            tile[(ra * RG + g0) * 64u] = ua;
        } else {
            T::put(ua, q0 % T::RPG, n0);
            T::put(ub, q1 % T::RPG, n1);
            tile[(ra * RG + g0) * 64u] = ua;
            tile[(ra * RG + g1) * 64u] = ub;
        }
        bad_a = (bad_a & ~((uint64_t)1 << q0)) | ((uint64_t)(n0 != T::ident(q0, N)) << q0);
        bad_a = (bad_a & ~((uint64_t)1 << q1)) | ((uint64_t)(n1 != T::ident(q1, N)) << q1);
    }

    // ---- the inverse: V <- V G (column operation on every row word), and its row mask -----------------------------------------------------
    const W m_cx = is_cx ? (W)1 : (W)0, m_sw = is_swap ? (W)1 : (W)0;
    uint32_t bv_lo = 0, bv_hi = 0;
#pragma unroll
    for (int k = 0; k < GPL; ++k) {
        const uint32_t g = j + (uint32_t)L * (uint32_t)k;
#pragma unroll
        for (uint32_t e = 0; e < T::RPG; ++e) {
            W w = T::get(v[k], e);
            w ^= ((w >> q1) & m_cx) << q0;                         // cx(q0, q1): column q0 ^= column q1
            const W x = ((w >> q0) ^ (w >> q1)) & m_sw;            // swap: columns q0, q1 trade places
            w ^= (x << q0) | (x << q1);
            T::put(v[k], e, w);
            const uint32_t r = T::RPG * g + e;
            const uint32_t differs = (uint32_t)(g < RG && w != T::ident(r, N));
            if (r < 32u) bv_lo |= differs << r;
            else bv_hi |= differs << (r - 32u);
        }
        if (g < RG && (is_cx || is_swap)) tile[(rv * RG + g) * 64u] = v[k];
    }
    bv_lo = lanes_or<L>(bv_lo);
    if (W64) bv_hi = lanes_or<L>(bv_hi);
    const uint64_t bad_v = (uint64_t)bv_lo | ((uint64_t)bv_hi << 32);

    if (j != 0) return;  // the scalars belong to lane 0
    if (a.flags & F_TRACK) {  // linear_function.rs:315-321: pushed whether or not the action is valid
        if ((uint32_t)(nf + nb) < a.sol_cap) {
            sol_at(a, env, (uint32_t)(nf + nb)) = sol_word_framed(act, iflags & LFD_FLAG_INVERTED);
            if (iflags & LFD_FLAG_INVERTED) ++nb;
            else ++nf;
        } else {
            fault |= 8u;
        }
    }
    depth = depth > 0 ? depth - 1 : 0;  // :323
    if ((a.flags & F_INVERTS) && (coin & 1u)) {  // maybe_random_invert (:227-235): the two matrices trade roles
        if (iflags & LFD_FLAG_SINGULAR) fault |= QG_FAULT_SINGULAR;  // `.expect("LFState is singular; cannot invert")`
        else iflags ^= LFD_FLAG_INVERTED;
    }
    // solved() (:325): the state is the identity iff its inverse is; read the mask of whichever region is the state now
    const bool swapped = (iflags & LFD_FLAG_INVERTED) != ra;
    const bool solved = (swapped ? bad_v : bad_a) == 0;
    const float achieved = solved ? 1.0f : 0.0f;
    const float reward = achieved - penalty;  // :326-327
    if (a.rewards_seq) a.rewards_seq[env] = reward;
    if (a.dones_seq) a.dones_seq[env] = (uint8_t)(depth == 0 || solved);
    bad2[ra] = bad_a;
    bad2[rv] = bad_v;
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);  // is_final (:334)
    a.success[env] = (uint8_t)solved;
    a.inverted[env] = (uint8_t)iflags;
    if (a.flags & F_TRACK) {
        a.sol_len[env * 2] = nf;
        a.sol_len[env * 2 + 1] = nb;
    }
    if (fault) atomicOr(&a.error[env], fault);
}

// ---- constructor state / set_state / reset / reset_done: thread per env, both matrices in LDS ([row][lane]: any per-lane row is conflict-free) ----
template <bool W64>
__global__ __launch_bounds__(64) void lfd_init_kernel(InitArgs a, uint32_t RG, const uint32_t *descs) {
    using T = LfdT<W64>;
    using W = typename T::W;
    extern __shared__ uint64_t lfd_lds_raw[];
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t L = threadIdx.x;
    if (env >= a.B) return;
    if (a.only_done && !a.done[env]) return;  // qg_vec_reset_done
    const uint32_t N = a.N, NP = RG * T::RPG;
    W *A = reinterpret_cast<W *>(lfd_lds_raw), *V = A + (size_t)NP * 64u;  // A[r * 64 + L], V[r * 64 + L]
    for (uint32_t r = 0; r < NP; ++r) A[r * 64u + L] = V[r * 64u + L] = T::ident(r, N);
    uint32_t flags = 0;
    auto row_op = [&](W *M, uint32_t desc) {  // LFState::cx / swap as a row operation on M
        const uint32_t kind = desc & 0xFFu, q0 = (desc >> 8) & 0xFFu, q1 = (desc >> 16) & 0xFFu;
        if (q0 == q1) return;
        const W r0 = M[q0 * 64u + L], r1 = M[q1 * 64u + L];
        if (kind == QG_CX) M[q1 * 64u + L] = r1 ^ r0;
        else if (kind == QG_SWAP) { M[q0 * 64u + L] = r1; M[q1 * 64u + L] = r0; }
    };
    if (a.mode == 1) {  // set_state (linear_function.rs:279-283): data[i] = x > 0
        for (uint32_t r = 0; r < N; ++r) {
            W w = 0;
            if (a.format == QG_FMT_PACKED) {
                w = reinterpret_cast<const W *>(a.src)[env * a.src_stride + r];
                if (N < 8 * sizeof(W)) w &= ((W)1 << N) - 1;
            } else if (a.format == QG_FMT_BITS) {  // the entry stream as bits (pack_bitstream)
                w = (W)bits_window(reinterpret_cast<const uint64_t *>(a.src), env * a.src_stride + (uint64_t)r * N, N);
            } else if (a.format == QG_FMT_I64) {
                const int64_t *p = reinterpret_cast<const int64_t *>(a.src) + env * a.src_stride + (uint64_t)r * N;
                for (uint32_t c = 0; c < N; ++c) w |= (W)(p[c] > 0) << c;
            } else {
                const int8_t *p = reinterpret_cast<const int8_t *>(a.src) + env * a.src_stride + (uint64_t)r * N;
                for (uint32_t c = 0; c < N; ++c) w |= (W)(p[c] > 0) << c;
            }
            A[r * 64u + L] = w;
        }
    } else if (a.mode == 2) {  // reset (:285-300): `difficulty` random gates on the identity; V = G_1 G_2 ... G_k: the same gates, last one first
        const uint64_t seed = init_seed(a);
        auto draw = [&](uint32_t t) -> int64_t {
            return a.actions ? (int64_t)a.actions[(uint64_t)t * a.B + env] : (int64_t)rng_action(seed, a.env_base + env, t, a.num_actions);
        };
        for (uint32_t t = 0; t < a.n_draws; ++t) {
            const int64_t act = draw(t);
            if (act >= 0 && act < (int64_t)a.num_actions) row_op(A, descs[act]);
        }
        for (uint32_t t = a.n_draws; t-- > 0;) {
            const int64_t act = draw(t);
            if (act >= 0 && act < (int64_t)a.num_actions) row_op(V, descs[act]);
        }
    }
    // both regions to memory: region 0 is the state (inverted := false, reset_internals :245-256)
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(2u * RG * 64u) + (env & 63u);
    auto store_region = [&](const W *M, uint32_t region) -> uint64_t {
        uint64_t bad = 0;
        for (uint32_t g = 0; g < RG; ++g) {
            uint4 u = make_uint4(0u, 0u, 0u, 0u);
            for (uint32_t e = 0; e < T::RPG; ++e) {
                const uint32_t r = T::RPG * g + e;
                const W w = M[r * 64u + L];
                T::put(u, e, w);
                bad |= (uint64_t)(w != T::ident(r, N)) << r;
            }
            tile[(region * RG + g) * 64u] = u;
        }
        return bad;
    };
    const uint64_t bad_a = store_region(A, 0);
    if (a.mode == 1) {  // V = A^-1 by the reference's Gauss-Jordan (:124-146), on the LDS copy of A
        for (uint32_t r = 0; r < NP; ++r) V[r * 64u + L] = T::ident(r, N);
        bool singular = false;
        for (uint32_t col = 0; col < N && !singular; ++col) {
            if (!((A[col * 64u + L] >> col) & 1)) {
                uint32_t p = col + 1;
                while (p < N && !((A[p * 64u + L] >> col) & 1)) ++p;
                if (p == N) { singular = true; break; }
                const W ta = A[col * 64u + L], tv = V[col * 64u + L];
                A[col * 64u + L] = A[p * 64u + L]; V[col * 64u + L] = V[p * 64u + L];
                A[p * 64u + L] = ta; V[p * 64u + L] = tv;
            }
            const W pa = A[col * 64u + L], pv = V[col * 64u + L];
            for (uint32_t r = 0; r < N; ++r)
                if (r != col && ((A[r * 64u + L] >> col) & 1)) { A[r * 64u + L] ^= pa; V[r * 64u + L] ^= pv; }
        }
        if (singular) flags |= LFD_FLAG_SINGULAR;
    }
    const uint64_t bad_v = store_region(V, 1);
    uint64_t *bad2 = reinterpret_cast<uint64_t *>(a.bad) + env * 2;
    bad2[0] = bad_a;
    bad2[1] = bad_v;
    const bool solved = bad_a == 0;
    a.depth[env] = a.depth_value;
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
    a.inverted[env] = (uint8_t)flags;
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

// ---- observe / get_state: one thread per (env, row) of the state's region ----------------------------------------------------------
template <bool W64>
__global__ __launch_bounds__(256) void lfd_export_kernel(ObsArgs a, uint32_t RG, const uint8_t *inverted) {
    using T = LfdT<W64>;
    using W = typename T::W;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid / a.N;
    if (env >= a.B) return;
    const uint32_t row = (uint32_t)(gid - env * a.N), N = a.N;
    const uint32_t region = inverted[env] & LFD_FLAG_INVERTED;
    const uint4 *tile = reinterpret_cast<const uint4 *>(a.state) + (env >> 6) * (uint64_t)(2u * RG * 64u) + (env & 63u);
    const W w = T::get(tile[(region * RG + row / T::RPG) * 64u], row % T::RPG);
    if (a.format == QG_FMT_PACKED) {
        reinterpret_cast<W *>(a.out)[env * a.out_stride + row] = w;
    } else if (a.format == QG_FMT_I64) {
        int64_t *o = reinterpret_cast<int64_t *>(a.out) + env * a.out_stride + (uint64_t)row * N;
        for (uint32_t c = 0; c < N; ++c) o[c] = (int64_t)((w >> c) & 1);
    } else {
        int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride + (uint64_t)row * N;
        for (uint32_t c = 0; c < N; ++c) o[c] = (int8_t)((w >> c) & 1);
    }
}

#ifndef LFD_LANES32
#define LFD_LANES32 1
#endif
#ifndef LFD_LANES64
#define LFD_LANES64 4
#endif

template <bool W64, int L>
static hipError_t lfd_launch_step(const StepArgs &a, uint32_t RG, hipStream_t s) {
    const dim3 grid(grid_for((uint64_t)L * a.B, 256)), block(256);
    const uint32_t gpl = (RG + (uint32_t)L - 1u) / (uint32_t)L;
#define LFD_CASE(G) case G: if constexpr (G * L <= 32 + L - 1) hipLaunchKernelGGL((lfd_step_kernel<W64, L, G>), grid, block, 0, s, a, RG); break;
    switch (gpl) {
        LFD_CASE(1) LFD_CASE(2) LFD_CASE(3) LFD_CASE(4) LFD_CASE(5) LFD_CASE(6) LFD_CASE(7) LFD_CASE(8)
        LFD_CASE(9) LFD_CASE(10) LFD_CASE(11) LFD_CASE(12) LFD_CASE(13) LFD_CASE(14) LFD_CASE(15) LFD_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef LFD_CASE
    return hipGetLastError();
}

// lanes per env: uint32 rows (N <= 32: <= 8 groups per matrix) / uint64 rows (<= 32 groups)
hipError_t lfd_step(const StepArgs &a, bool w64, uint32_t RG, hipStream_t s) {
    if (!a.B) return hipSuccess;
    if (a.T != 1) return hipErrorInvalidValue;  // the host issues T single steps
    return w64 ? lfd_launch_step<true, LFD_LANES64>(a, RG, s) : lfd_launch_step<false, LFD_LANES32>(a, RG, s);
}
hipError_t lfd_init(const InitArgs &a, bool w64, uint32_t RG, const uint32_t *descs, hipStream_t s) {
    if (!a.B) return hipSuccess;
    const size_t lds = (size_t)2 * RG * (w64 ? 2 : 4) * (w64 ? 8 : 4) * 64;  // two matrices of NP words per lane
    if (w64) hipLaunchKernelGGL(lfd_init_kernel<true>, dim3(grid_for(a.B, 64)), dim3(64), lds, s, a, RG, descs);
    else hipLaunchKernelGGL(lfd_init_kernel<false>, dim3(grid_for(a.B, 64)), dim3(64), lds, s, a, RG, descs);
    return hipGetLastError();
}
hipError_t lfd_export(const ObsArgs &a, bool w64, uint32_t RG, const uint8_t *inverted, hipStream_t s) {
    if (!a.B) return hipSuccess;
    const dim3 grid(grid_for(a.B * a.N, 256)), block(256);
    if (w64) hipLaunchKernelGGL(lfd_export_kernel<true>, grid, block, 0, s, a, RG, inverted);
    else hipLaunchKernelGGL(lfd_export_kernel<false>, grid, block, 0, s, a, RG, inverted);
    return hipGetLastError();
}

}  // namespace qg
