// kernels_small.hip -- one-word-per-env kernels (thread per env):
//   LF8  LinearFunctionEnv with N <= 8: the N x N GF(2) matrix is one uint64, byte r = row r
//        (rust/src/envs/linear_function.rs:29-151, step :302-328)
//   PERM PermutationEnv with N <= 16: nibble i of one uint64 = state[i]
//        (rust/src/envs/permutation.rs:101-128, step :194-225)
// Loads and stores are one coalesced 8 B access per lane (512 B per wave instruction).
#include "device_common.hpp"

namespace qg {

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

__host__ __device__ inline uint64_t lf8_identity(uint32_t N) {
    uint64_t id = 0;
    for (uint32_t r = 0; r < N; ++r) id |= (1ull << r) << (8 * r);
    return id;
}
__host__ __device__ inline uint64_t perm_identity(uint32_t N) {
    uint64_t id = 0;
    for (uint32_t i = 0; i < N; ++i) id |= (uint64_t)i << (4 * i);
    return id;
}

// LFState::cx / swap (linear_function.rs:62-83); other gate kinds compile to OP_NONE (:241)
__device__ inline uint64_t lf8_apply(uint64_t s, uint32_t ops) {
    const uint32_t op = ops & 0x3FFFu, type = op >> 12, dst = op & 63u, src = (op >> 6) & 63u;
    const uint64_t a = (s >> (8 * src)) & 0xFFull, b = (s >> (8 * dst)) & 0xFFull;
    if (type == OP_XOR) s ^= a << (8 * dst);
    if (type == OP_SWAP) s ^= ((a ^ b) << (8 * src)) | ((a ^ b) << (8 * dst));
    return s;
}
// LFState::inverse (linear_function.rs:124-146) on the byte-packed word; false if singular
__device__ inline bool lf8_inverse(uint64_t m, uint32_t N, uint64_t &out) {
    const uint64_t ones = 0x0101010101010101ull;
    uint64_t v = lf8_identity(N);
    bool ok = true;
    for (uint32_t col = 0; col < N; ++col) {
        uint64_t colbits = (m >> col) & ones;                 // byte r = bit (r, col)
        uint64_t cand = colbits & (~0ull << (8 * col));       // rows >= col
        if (!cand) { ok = false; continue; }
        uint32_t p = (uint32_t)(__ffsll((long long)cand) - 1) >> 3;
        // swap rows col <-> p in both
        uint64_t x = ((m >> (8 * col)) ^ (m >> (8 * p))) & 0xFFull;
        m ^= (x << (8 * col)) | (x << (8 * p));
        uint64_t y = ((v >> (8 * col)) ^ (v >> (8 * p))) & 0xFFull;
        v ^= (y << (8 * col)) | (y << (8 * p));
        const uint64_t pm = (m >> (8 * col)) & 0xFFull, pv = (v >> (8 * col)) & 0xFFull;
        uint64_t hit = ((m >> col) & ones) & ~(1ull << (8 * col));  // rows != col with bit col
        m ^= hit * pm;  // hit has 0/1 per byte and pm < 256: no carries between bytes
        v ^= hit * pv;
    }
    out = v;
    return ok;
}

__device__ inline uint64_t perm_apply(uint64_t s, uint32_t ops) {  // permutation.rs:205-208
    const uint32_t op = ops & 0x3FFFu, type = op >> 12, dst = op & 63u, src = (op >> 6) & 63u;
    if (type == OP_SWAP) {
        uint64_t x = ((s >> (4 * src)) ^ (s >> (4 * dst))) & 0xFull;
        s ^= (x << (4 * src)) | (x << (4 * dst));
    }
    return s;
}
__device__ inline uint64_t perm_invert(uint64_t s, uint32_t N) {  // permutation.rs:101-107
    uint64_t inv = 0;
    for (uint32_t i = 0; i < N; ++i) inv |= (uint64_t)i << (4 * ((s >> (4 * i)) & 0xFull));
    return inv;
}

// One env's step(s) (Env::step, linear_function.rs:302-328 / permutation.rs:194-225).  `fresh`: the env has just been reset in this launch
// (word_reset_step_kernel): its state and depth come in registers -- `fresh_state`, `fresh_depth` -- instead of from memory, its `inverted` flag is
// clear, and its state is stored whether or not the step changes it.
template <bool PERM>
__device__ inline void word_step_body(const StepArgs &a, uint64_t env, bool fresh = false, uint64_t fresh_state = 0, int32_t fresh_depth = 0) {
    const bool act64 = a.flags & F_ACT64;
    uint64_t *sp = reinterpret_cast<uint64_t *>(a.state) + env;
    uint64_t s = fresh ? fresh_state : *sp;
    const uint64_t s0 = fresh ? ~s : s;
    int32_t depth = fresh ? fresh_depth : a.depth[env];
    uint32_t inverted = ((a.flags & F_INVERTS) && !fresh) ? a.inverted[env] : 0u;
    const uint64_t ident = PERM ? perm_identity(a.N) : lf8_identity(a.N);
    bool solved = false;
    float reward = 0.0f;
    uint32_t fault = 0;

    for (uint32_t t = 0; t < a.T; ++t) {
        const int64_t act = load_action(a.actions, (uint64_t)t * a.B + env, act64);
        const bool in_range = act >= 0 && act < (int64_t)a.num_actions;
        GateEntry g = {0u, 0.0f};
        if (in_range) g = a.gates[act];
        float penalty = g.penalty;
        if ((a.flags & F_LAYERS) && in_range)
            penalty = layers_penalty(layer_rec(a.layers, env, 2 * a.N + 2), a.N, a.descs[act], a.w);
        s = PERM ? perm_apply(s, g.ops) : lf8_apply(s, g.ops);

        // solution push: Permutation only for a valid action (permutation.rs:210-216),
        // LinearFunction always (linear_function.rs:315-321)
        if ((a.flags & F_TRACK) && (!PERM || in_range)) {
            int32_t nf = a.sol_len[env * 2], nb = a.sol_len[env * 2 + 1];
            if ((uint32_t)(nf + nb) < a.sol_cap) {
                sol_at(a, env, (uint32_t)(nf + nb)) = sol_word_framed(act, inverted);
                a.sol_len[env * 2 + (inverted ? 1 : 0)] = (inverted ? nb : nf) + 1;
            } else {
                fault |= 8u;
            }
        }
        // (Permutation inverts before the depth decrement, permutation.rs:219-221; the two
        // updates are independent so the order is unobservable.)
        depth = depth > 0 ? depth - 1 : 0;
        if (a.flags & F_INVERTS) {
            uint32_t coin = a.coins ? a.coins[(uint64_t)t * a.B + env]
                                    : (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a) + t) >> 63);
            if (coin & 1u) {
                if (PERM) {
                    s = perm_invert(s, a.N);
                    inverted ^= 1u;
                } else {
                    uint64_t inv;
                    if (lf8_inverse(s, a.N, inv)) {
                        s = inv;
                        inverted ^= 1u;
                    } else {
                        fault |= QG_FAULT_SINGULAR;
                    }
                }
            }
        }
        solved = (s == ident);
        const float achieved = solved ? 1.0f : 0.0f;
        reward = achieved - penalty;
        if (a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
        if (a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
    }
    if (s != s0) *sp = s;
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (a.flags & F_INVERTS) a.inverted[env] = (uint8_t)inverted;
    if (fault) atomicOr(&a.error[env], fault);
}

template <bool PERM>
__global__ __launch_bounds__(256) void word_step_kernel(StepArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    QG_PREFETCH_STEP_ARGS(a);  // qgym_internal.hpp
    if (env >= a.B) return;
    word_step_body<PERM>(a, env);
}

// ---- Env::reset of a few finished envs scattered over the batch (qg_vec_reset_done, qg_vec_reset_done_step) ---------------------------------
// A scramble is `difficulty` row operations (position swaps) applied to the identity one after the other: a chain of dependent updates, each
// behind a gate-table load -- 64 draws cost a lone lane ~30 us, whatever the other 63 lanes of its wave do.  The updates compose: for
// LinearFunctionEnv a run of row operations IS the matrix E it makes of the identity, and a later run after an earlier one is the GF(2)
// product E_later . E_earlier; for PermutationEnv a run of position swaps applied to x gives x[e[i]] where e is what the run makes of the
// identity, so later after earlier is earlier[later[i]].  So a finished env is given 16 lanes: each applies its ceil(difficulty / 16)
// consecutive draws to the identity (all its gate entries requested at once), and four rounds of products over the 16 lanes leave the
// env's fresh state on all of them -- one memory round trip and ~300 instructions instead of `difficulty` dependent round trips.  A wave
// takes four finished envs at a time.
// E_hi . E_lo on byte-packed rows: row r = xor over c of hi[r][c] * (row c of lo)
__device__ inline uint64_t lf8_mul(uint64_t hi, uint64_t lo) {
    const uint32_t ones = 0x01010101u;
    const uint32_t h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32), l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32);
    uint32_t c0 = 0, c1 = 0;
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) {
        const uint32_t row = __builtin_amdgcn_perm(0u, c < 4 ? l0 : l1, 0x01010101u * (c & 3u));  // row c of lo in every byte
        const uint32_t m0 = (h0 >> c) & ones, m1 = (h1 >> c) & ones;                              // byte r = hi[r][c]
        c0 ^= ((m0 << 8) - m0) & row;  // m * 255: 0xFF in the bytes of the rows that take row c
        c1 ^= ((m1 << 8) - m1) & row;
    }
    return (uint64_t)c0 | ((uint64_t)c1 << 32);
}
// positions moved by `hi` after `lo`: out[i] = lo[hi[i]] (nibbles)
__device__ inline uint64_t perm_compose(uint64_t hi, uint64_t lo, uint32_t N) {
    uint64_t out = 0;
    for (uint32_t i = 0; i < N; ++i) out |= ((lo >> (4u * (uint32_t)((hi >> (4u * i)) & 0xFull))) & 0xFull) << (4u * i);
    return out;
}
constexpr uint32_t WORD_COOP_MAX = 24;  // finished envs per wave the 16-lane groups take, four per pass (~1.8 us a pass); a fuller wave runs the per-lane chain (~15 us for 64 draws)

// Bit l of `m` (wave-uniform): lane l's env starts over, its draws from the counter RNG.  Call from all 64 lanes; returns the env's fresh state
// on the lanes of `m` (the identity elsewhere).  `env0`: the env of lane 0.
template <bool PERM>
__device__ inline uint64_t word_scramble_coop(const InitArgs &a, uint64_t env0, uint64_t m, uint64_t ident) {
    const uint32_t lane = __lane_id(), grp = lane >> 4, j = lane & 15u;
    const uint32_t per = (a.n_draws + 15u) >> 4;  // consecutive draws per lane
    const uint64_t seed = init_seed(a);
    const uint32_t my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));  // (on the lanes of m) which finished env of the wave this one is
    uint64_t mine = ident, rest = m;
    for (uint32_t pass = 0; rest; ++pass) {
        uint32_t src = 64;  // the lane whose env this group scrambles: the grp-th set bit of `rest`
#pragma unroll
        for (uint32_t g = 0; g < 4; ++g) {
            if (rest) {
                if (g == grp) src = (uint32_t)__ffsll((long long)rest) - 1u;
                rest &= rest - 1ull;
            }
        }
        uint64_t s = ident;
        if (src < 64u) {
            const uint64_t e = a.env_base + env0 + src;
            const uint32_t t0 = j * per, t1 = t0 + per < a.n_draws ? t0 + per : a.n_draws;
            for (uint32_t t = t0; t < t1; t += 4u) {
                uint32_t ops[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) ops[k] = t + k < t1 ? a.gates[rng_action(seed, e, t + k, a.num_actions)].ops : 0u;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) s = PERM ? perm_apply(s, ops[k]) : lf8_apply(s, ops[k]);
            }
        }
#pragma unroll
        for (uint32_t d = 1; d < 16u; d <<= 1) {  // the lane with bit d set holds the LATER draws
            const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)s, (int)d);
            const uint64_t hi = (j & d) ? s : o, lo = (j & d) ? o : s;
            s = PERM ? perm_compose(hi, lo, a.N) : lf8_mul(hi, lo);
        }
        const uint64_t got = (uint64_t)__shfl((unsigned long long)s, (int)((my_rank & 3u) << 4));  // from the group that took this lane's env
        if (((m >> lane) & 1ull) && (my_rank >> 2) == pass) mine = got;
    }
    return mine;
}

// the per-lane chain (full resets, reset_with's given draws, waves full of finished envs): eight gate entries requested ahead of the updates
template <bool PERM>
__device__ inline uint64_t word_scramble_lane(const InitArgs &a, uint64_t env, uint64_t s) {
    const uint64_t seed = init_seed(a);
    for (uint32_t t = 0; t < a.n_draws; t += 8u) {
        uint32_t ops[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            ops[k] = 0u;
            if (t + k < a.n_draws) {
                const int64_t act = a.actions ? (int64_t)a.actions[(uint64_t)(t + k) * a.B + env] : (int64_t)rng_action(seed, a.env_base + env, t + k, a.num_actions);
                if (act >= 0 && act < (int64_t)a.num_actions) ops[k] = a.gates[act].ops;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) s = PERM ? perm_apply(s, ops[k]) : lf8_apply(s, ops[k]);
    }
    return s;
}

// what Env::reset / set_state leave besides the state (reset_internals, linear_function.rs:245-256): log lengths, fault word, layer record
__device__ inline void word_reset_bookkeeping(const InitArgs &a, uint64_t env, uint32_t fault) {
    a.inverted[env] = 0;
    a.error[env] = fault;
    a.sol_len[env * 2] = 0;
    a.sol_len[env * 2 + 1] = 0;
    if (a.layers) {
        const LayerRec lay = layer_rec(a.layers, env, a.layers_len);
        for (uint32_t i = 0; i + 2 < a.layers_len; ++i) lay[i] = -1;
        lay[a.layers_len - 2] = 0;
        lay[a.layers_len - 1] = 0;
    }
}

template <bool PERM>
__global__ __launch_bounds__(256) void word_init_kernel(InitArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = env < a.B && !(a.only_done && !a.done[env]);  // (only_done: qg_vec_reset_done)
    const uint64_t ident = PERM ? perm_identity(a.N) : lf8_identity(a.N);
    uint64_t s = ident;
    uint32_t fault = 0;
    bool scrambled = false;
    if (a.mode == 2 && a.only_done && !a.actions) {  // (wave-uniform) a few finished envs per wave: 16 lanes each
        const uint64_t m = __ballot(mine);
        if (m && (uint32_t)__popcll(m) <= WORD_COOP_MAX) {
            s = word_scramble_coop<PERM>(a, env - __lane_id(), m, ident);
            scrambled = true;
        }
    }
    if (!mine) return;
    if (a.mode == 1) {
        s = 0;
        if (PERM) {  // permutation.rs:168-173: state[i] = x as usize
            for (uint32_t i = 0; i < a.N; ++i) {
                int64_t v;
                if (a.format == QG_FMT_I64) v = reinterpret_cast<const int64_t *>(a.src)[env * a.src_stride + i];
                else v = reinterpret_cast<const uint8_t *>(a.src)[env * a.src_stride + i];  // U8 / PACKED: one byte per entry
                if (v < 0 || v >= (int64_t)a.N) { fault |= QG_FAULT_BAD_STATE; v = 0; }
                s |= (uint64_t)v << (4 * i);
            }
            if (a.inverts) {  // a repeated entry has no inverse (perm_invert): a fault with add_inverts only, like kernels_perm.hip
                uint32_t seen = 0;
                for (uint32_t i = 0; i < a.N; ++i) {
                    const uint32_t bit = 1u << ((s >> (4 * i)) & 0xFull);
                    if (seen & bit) fault |= QG_FAULT_BAD_STATE;
                    seen |= bit;
                }
            }
        } else {  // linear_function.rs:279-283
            for (uint32_t r = 0; r < a.N; ++r) {
                uint64_t w = 0;
                if (a.format == QG_FMT_PACKED) {
                    w = reinterpret_cast<const uint32_t *>(a.src)[env * a.src_stride + r] & ((1u << a.N) - 1u);
                } else if (a.format == QG_FMT_I64) {
                    const int64_t *p = reinterpret_cast<const int64_t *>(a.src) + env * a.src_stride + (uint64_t)r * a.N;
                    for (uint32_t c = 0; c < a.N; ++c) w |= (uint64_t)(p[c] > 0) << c;
                } else {
                    const int8_t *p = reinterpret_cast<const int8_t *>(a.src) + env * a.src_stride + (uint64_t)r * a.N;
                    for (uint32_t c = 0; c < a.N; ++c) w |= (uint64_t)(p[c] > 0) << c;
                }
                s |= w << (8 * r);
            }
        }
    } else if (a.mode == 2 && !scrambled) {
        s = word_scramble_lane<PERM>(a, env, s);
    }
    const bool solved = (s == ident);
    reinterpret_cast<uint64_t *>(a.state)[env] = s;
    a.depth[env] = a.depth_value;
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
    word_reset_bookkeeping(a, env, fault);
}

// qg_vec_reset_done_step in one launch: the envs whose episode is over (`done`, as the previous step left it) start over -- 16 lanes each, or the
// per-lane chain in a wave full of them -- and then EVERY env takes its step; a fresh env's state and depth reach its step in registers.
template <bool PERM>
__global__ __launch_bounds__(256) void word_reset_step_kernel(InitArgs ia, StepArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    QG_PREFETCH_STEP_ARGS(a);  // qgym_internal.hpp
    const bool fin = env < a.B && ia.done[env];
    const uint64_t m = __ballot(fin);
    const uint64_t ident = PERM ? perm_identity(a.N) : lf8_identity(a.N);
    uint64_t fresh = ident;
    if (m) {  // (wave-uniform)
        if ((uint32_t)__popcll(m) <= WORD_COOP_MAX) fresh = word_scramble_coop<PERM>(ia, env - __lane_id(), m, ident);
        else if (fin) fresh = word_scramble_lane<PERM>(ia, env, ident);
    }
    if (env >= a.B) return;
    if (fin) word_reset_bookkeeping(ia, env, 0u);  // (read back by this same lane's step where the step reads them at all)
    word_step_body<PERM>(a, env, fin, fresh, ia.depth_value);
}

template <bool PERM>
__global__ __launch_bounds__(256) void word_export_kernel(ObsArgs a) {
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.B) return;
    const uint64_t s = reinterpret_cast<const uint64_t *>(a.state)[env];
    const uint32_t N = a.N;
    if (PERM) {
        if (a.format == QG_FMT_I64) {  // get_state (permutation.rs:130-132)
            int64_t *o = reinterpret_cast<int64_t *>(a.out) + env * a.out_stride;
            for (uint32_t i = 0; i < N; ++i) o[i] = (int64_t)((s >> (4 * i)) & 0xFull);
        } else if (a.format == QG_FMT_PACKED) {
            uint8_t *o = reinterpret_cast<uint8_t *>(a.out) + env * a.out_stride;
            for (uint32_t i = 0; i < N; ++i) o[i] = (uint8_t)((s >> (4 * i)) & 0xFull);
        } else {  // observe: indices i*N + state[i] (permutation.rs:241-243), densified
            int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride;
            for (uint32_t i = 0; i < N; ++i) {
                uint32_t v = (uint32_t)((s >> (4 * i)) & 0xFull);
                for (uint32_t c = 0; c < N; ++c) o[i * N + c] = (int8_t)(c == v);
            }
        }
    } else {
        if (a.format == QG_FMT_PACKED) {
            uint32_t *o = reinterpret_cast<uint32_t *>(a.out) + env * a.out_stride;
            for (uint32_t r = 0; r < N; ++r) o[r] = (uint32_t)((s >> (8 * r)) & 0xFFull);
        } else if (a.format == QG_FMT_I64) {
            int64_t *o = reinterpret_cast<int64_t *>(a.out) + env * a.out_stride;
            for (uint32_t r = 0; r < N; ++r)
                for (uint32_t c = 0; c < N; ++c) o[r * N + c] = (int64_t)((s >> (8 * r + c)) & 1ull);
        } else {
            int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride;
            for (uint32_t r = 0; r < N; ++r)
                for (uint32_t c = 0; c < N; ++c) o[r * N + c] = (int8_t)((s >> (8 * r + c)) & 1ull);
        }
    }
}

hipError_t lf8_step(const StepArgs &a, bool, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(word_step_kernel<false>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t perm_step(const StepArgs &a, bool, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(word_step_kernel<true>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t lf8_init(const InitArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(word_init_kernel<false>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t perm_init(const InitArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(word_init_kernel<true>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t word_reset_step(const InitArgs &reset, const StepArgs &a, bool perm, hipStream_t s) {
    if (!a.B) return hipSuccess;
    if (perm) hipLaunchKernelGGL(word_reset_step_kernel<true>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, reset, a);
    else hipLaunchKernelGGL(word_reset_step_kernel<false>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, reset, a);
    return hipGetLastError();
}
hipError_t lf8_export(const ObsArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(word_export_kernel<false>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t perm_export(const ObsArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(word_export_kernel<true>, dim3(grid_for(a.B, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace qg
