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

template <bool PERM>
__global__ __launch_bounds__(256) void word_step_kernel(StepArgs a) {
    KernelClock kclk(a.kclk, a.kclk_waves);  // device_common.hpp
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    QG_PREFETCH_STEP_ARGS(a);  // qgym_internal.hpp
    if (env >= a.B) return;
    const bool act64 = a.flags & F_ACT64;
    uint64_t *sp = reinterpret_cast<uint64_t *>(a.state) + env;
    uint64_t s = *sp;
    const uint64_t s0 = s;
    int32_t depth = a.depth[env];
    uint32_t inverted = (a.flags & F_INVERTS) ? a.inverted[env] : 0u;
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
__global__ __launch_bounds__(256) void word_init_kernel(InitArgs a) {
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.B) return;
    if (a.only_done && !a.done[env]) return;  // qg_vec_reset_done
    const uint64_t ident = PERM ? perm_identity(a.N) : lf8_identity(a.N);
    uint64_t s = ident;
    uint32_t fault = 0;
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
    } else if (a.mode == 2) {
        for (uint32_t t = 0; t < a.n_draws; ++t) {
            int64_t act = a.actions ? (int64_t)a.actions[(uint64_t)t * a.B + env]
                                    : (int64_t)rng_action(init_seed(a), a.env_base + env, t, a.num_actions);
            uint32_t ops = (act >= 0 && act < (int64_t)a.num_actions) ? a.gates[act].ops : 0u;
            s = PERM ? perm_apply(s, ops) : lf8_apply(s, ops);
        }
    }
    const bool solved = (s == ident);
    reinterpret_cast<uint64_t *>(a.state)[env] = s;
    a.depth[env] = a.depth_value;
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
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
