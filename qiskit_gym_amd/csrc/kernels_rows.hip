// kernels_rows.hip -- ROWS-layout kernels: CliffordEnv (any N <= 32) and LinearFunctionEnv N > 8.
//
// Reference semantics implemented here (paths relative to the reference repo):
//   Clifford::step              rust/src/envs/clifford.rs:321-347
//   CFState row ops / gates     rust/src/envs/clifford.rs:64-133
//   CFState::solved             rust/src/envs/clifford.rs:136-145
//   CFState::inverse            rust/src/envs/clifford.rs:147-170 (Gauss-Jordan, first pivot below)
//   maybe_random_invert         rust/src/envs/clifford.rs:262-270
//   LinearFunction::step        rust/src/envs/linear_function.rs:302-328 (same skeleton)
//   set_state / reset           rust/src/envs/clifford.rs:299-319
//   observe + Gym densify       rust/src/envs/clifford.rs:361-368, src/qiskit_gym/envs/adapters.py:50-54
//
// Mapping to the machine: a lane owns 16 B = RPL consecutive rows of one env; L lanes (a power of
// two) form an env, so one wavefront holds 64/L envs and its state load/store is a single fully
// coalesced 16 B/lane access.  A row operation fetches the source row from its owner lane with a
// ds_bpermute shuffle; the identity test is a per-lane compare against constant unit rows folded
// across the env's lanes with one wave ballot.  No MFMA: this is GF(2) row arithmetic.
#include "device_common.hpp"

namespace qg {

template <typename W>
struct RowTraits;
template <>
struct RowTraits<uint32_t> {
    static constexpr int RPL = 4;
};
template <>
struct RowTraits<uint64_t> {
    static constexpr int RPL = 2;
};

template <typename W>
__device__ inline void load_rows(const void *p, W (&r)[RowTraits<W>::RPL]) {
    uint4 q = *reinterpret_cast<const uint4 *>(p);
    if constexpr (sizeof(W) == 4) {
        r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
    } else {
        r[0] = (uint64_t)q.x | ((uint64_t)q.y << 32);
        r[1] = (uint64_t)q.z | ((uint64_t)q.w << 32);
    }
}
template <typename W>
__device__ inline void store_rows(void *p, const W (&r)[RowTraits<W>::RPL]) {
    uint4 q;
    if constexpr (sizeof(W) == 4) {
        q.x = r[0]; q.y = r[1]; q.z = r[2]; q.w = r[3];
    } else {
        q.x = (uint32_t)r[0]; q.y = (uint32_t)(r[0] >> 32);
        q.z = (uint32_t)r[1]; q.w = (uint32_t)(r[1] >> 32);
    }
    *reinterpret_cast<uint4 *>(p) = q;
}
template <typename W>
__device__ inline W sel(const W (&r)[RowTraits<W>::RPL], uint32_t c) {
    if constexpr (sizeof(W) == 4) {
        W lo = (c & 1) ? r[1] : r[0];
        W hi = (c & 1) ? r[3] : r[2];
        return (c & 2) ? hi : lo;
    } else {
        return (c & 1) ? r[1] : r[0];
    }
}
template <typename W>
__device__ inline void put(W (&r)[RowTraits<W>::RPL], uint32_t c, W v) {
#pragma unroll
    for (int i = 0; i < RowTraits<W>::RPL; ++i) r[i] = (c == (uint32_t)i) ? v : r[i];
}
template <typename W>
__device__ inline W shfl_w(W v, uint32_t src_lane) {
    if constexpr (sizeof(W) == 4) {
        return (W)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
    } else {
        uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)v);
        uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)(v >> 32));
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
}
__device__ inline uint32_t shfl_u32(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

// unit rows this lane must hold when the env is solved (clifford.rs:136-145); pad rows are zero
template <typename W>
__device__ inline void identity_rows(W (&id)[RowTraits<W>::RPL], uint32_t lie, uint32_t D) {
#pragma unroll
    for (int c = 0; c < RowTraits<W>::RPL; ++c) {
        uint32_t row = lie * RowTraits<W>::RPL + c;
        id[c] = row < D ? (W)1 << row : (W)0;
    }
}

// Apply the (<= 2, disjoint) row operations of one action.  Returns true if this lane's rows
// were written.  `base` = first lane of this env inside the wave.
template <typename W>
__device__ inline bool apply_ops(W (&r)[RowTraits<W>::RPL], uint32_t ops, uint32_t lie, uint32_t base) {
    constexpr uint32_t RPL = RowTraits<W>::RPL;
    bool wrote = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        uint32_t op = (ops >> (14 * k)) & 0x3FFFu;
        uint32_t type = op >> 12, dst = op & 63u, src = (op >> 6) & 63u;
        // every lane executes the shuffles (other envs in the wave may need them)
        W sv = shfl_w<W>(sel<W>(r, src % RPL), base + src / RPL);
        W dv = shfl_w<W>(sel<W>(r, dst % RPL), base + dst / RPL);
        bool own_dst = (type != OP_NONE) && (dst / RPL == lie);
        bool own_src = (type == OP_SWAP) && (src / RPL == lie);
        W nd = (type == OP_SWAP) ? sv : (W)(dv ^ sv);  // row_xor (clifford.rs:64-72) / swap_rows (:74-82)
        if (own_dst) put<W>(r, dst % RPL, nd);
        if (own_src) put<W>(r, src % RPL, dv);
        wrote |= own_dst | own_src;
    }
    return wrote;
}

// Gauss-Jordan inverse over GF(2) of the env spread over `L` lanes (clifford.rs:147-170).
// m: this lane's rows of the matrix (consumed); v: receives this lane's rows of the inverse.
// Returns false (uniformly per env) when the matrix is singular -- the reference panics there.
template <typename W>
__device__ inline bool gf2_inverse(W (&m)[RowTraits<W>::RPL], W (&v)[RowTraits<W>::RPL], uint32_t D, uint32_t L,
                                   uint32_t lie, uint32_t base, uint32_t lane) {
    constexpr uint32_t RPL = RowTraits<W>::RPL;
    identity_rows<W>(v, lie, D);
    const uint64_t gmask = (L >= 64) ? ~0ull : (((1ull << L) - 1ull) << base);
    bool singular = false;
    for (uint32_t col = 0; col < D; ++col) {
        const uint32_t olane = base + col / RPL, ocomp = col % RPL;
        // first row >= col with bit `col` set: row `col` itself if its diagonal bit is set,
        // otherwise the reference's `((col+1)..dim).find(..)` (clifford.rs:153-155)
        uint32_t cand = RPL;
#pragma unroll
        for (int c = RPL - 1; c >= 0; --c) {
            uint32_t row = lie * RPL + c;
            if (row >= col && row < D && ((m[c] >> col) & 1)) cand = c;
        }
        uint64_t bal = __ballot(cand < RPL) & gmask;
        if (bal == 0) singular = true;  // uniform per env
        uint32_t plane = bal ? (uint32_t)(__ffsll((long long)bal) - 1) : olane;
        uint32_t pcomp = shfl_u32(cand, plane) & (RPL - 1);
        W pm = shfl_w<W>(sel<W>(m, pcomp), plane), pv = shfl_w<W>(sel<W>(v, pcomp), plane);
        W cm = shfl_w<W>(sel<W>(m, ocomp), olane), cv = shfl_w<W>(sel<W>(v, ocomp), olane);
        if (lane == plane) { put<W>(m, pcomp, cm); put<W>(v, pcomp, cv); }  // swap_rows(col, pivot)
        if (lane == olane) { put<W>(m, ocomp, pm); put<W>(v, ocomp, pv); }
#pragma unroll
        for (int c = 0; c < (int)RPL; ++c) {  // eliminate column `col` from every other row (:160-165)
            uint32_t row = lie * RPL + c;
            bool hit = (row != col) && ((m[c] >> col) & 1);
            m[c] ^= hit ? pm : (W)0;
            v[c] ^= hit ? pv : (W)0;
        }
    }
    return !singular;
}

// ------------------------------------------------------------------------------------------
// step / fused rollout
// ------------------------------------------------------------------------------------------
template <typename W>
__global__ __launch_bounds__(256) void rows_step_kernel(StepArgs a) {
    constexpr uint32_t RPL = RowTraits<W>::RPL;
    extern __shared__ GateEntry s_gates[];

    const uint32_t L = 1u << a.log2L;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env_raw = gid >> a.log2L;
    const uint32_t lie = (uint32_t)gid & (L - 1);
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    const uint32_t base = lane & ~(L - 1);
    const bool valid = env_raw < a.B;
    const uint64_t env = valid ? env_raw : a.B - 1;  // tail lanes shadow the last env, never store
    const bool leader = valid && lie == 0;
    const bool act64 = a.flags & F_ACT64;

    // issue the long-latency loads first
    char *sp = reinterpret_cast<char *>(a.state) + ((env << a.log2L) + lie) * 16;
    W r[RPL];
    load_rows<W>(sp, r);
    int64_t act = load_action(a.actions, env, act64);
    int32_t depth = a.depth[env];
    uint32_t inverted = (a.flags & F_INVERTS) ? a.inverted[env] : 0u;

    // stage the gate table in LDS while those are in flight
    const bool lds_table = a.num_actions * sizeof(GateEntry) <= 32768;
    if (lds_table) {
        for (uint32_t i = threadIdx.x; i < a.num_actions; i += blockDim.x) s_gates[i] = a.gates[i];
        __syncthreads();
    }

    W ident[RPL];
    identity_rows<W>(ident, lie, a.D);
    const uint64_t gmask = (L >= 64) ? ~0ull : (((1ull << L) - 1ull) << base);

    bool dirty = false;
    bool solved = false;
    float reward = 0.0f;
    uint32_t fault = 0;

    for (uint32_t t = 0; t < a.T; ++t) {
        if (t) act = load_action(a.actions, (uint64_t)t * a.B + env, act64);
        const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (clifford.rs:324)
        GateEntry g = {0u, 0.0f};
        if (in_range) g = lds_table ? s_gates[act] : a.gates[act];
        float penalty = g.penalty;
        if ((a.flags & F_LAYERS) && leader && in_range)
            penalty = layers_penalty(a.layers + env * (2 * a.N + 2), a.N, a.descs[act], a.w);

        dirty |= apply_ops<W>(r, g.ops, lie, base);  // apply_gate_to_state (clifford.rs:331)

        if ((a.flags & F_TRACK) && leader) {  // clifford.rs:334-340: pushed whether or not the action was valid
            int32_t nf = a.sol_len[env * 2], nb = a.sol_len[env * 2 + 1];
            if ((uint32_t)(nf + nb) < a.sol_cap) {
                sol_at(a, env, (uint32_t)(nf + nb)) = sol_word_framed(act, inverted);
                a.sol_len[env * 2 + (inverted ? 1 : 0)] = (inverted ? nb : nf) + 1;
            } else {
                fault |= 8u;
            }
        }

        depth = depth > 0 ? depth - 1 : 0;  // saturating_sub (clifford.rs:342)

        if (a.flags & F_INVERTS) {  // maybe_random_invert (clifford.rs:262-270)
            uint32_t coin = a.coins ? a.coins[(uint64_t)t * a.B + env]
                                    : (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a) + t) >> 63);
            coin = valid ? (coin & 1u) : 0u;
            if (__any((int)coin)) {
                W m[RPL], v[RPL];
#pragma unroll
                for (int c = 0; c < (int)RPL; ++c) m[c] = r[c];
                bool ok = gf2_inverse<W>(m, v, a.D, L, lie, base, lane);
                if (coin && ok) {
#pragma unroll
                    for (int c = 0; c < (int)RPL; ++c) r[c] = v[c];
                    dirty = true;
                    inverted ^= 1u;
                }
                if (coin && !ok) fault |= QG_FAULT_SINGULAR;
            }
        }

        bool ok = true;
#pragma unroll
        for (int c = 0; c < (int)RPL; ++c) ok &= (r[c] == ident[c]);
        const uint64_t bal = __ballot(ok);
        solved = (bal & gmask) == gmask;  // CFState::solved (clifford.rs:136-145)
        const float achieved = solved ? 1.0f : 0.0f;
        reward = achieved - penalty;  // clifford.rs:345-346

        if (leader) {
            if (a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
            if (a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
        }
    }

    if (valid && dirty) store_rows<W>(sp, r);
    if (leader) {
        a.depth[env] = depth;
        a.reward[env] = reward;
        a.done[env] = (uint8_t)(depth == 0 || solved);  // is_final (clifford.rs:353)
        a.success[env] = (uint8_t)solved;
        if (a.flags & F_INVERTS) a.inverted[env] = (uint8_t)inverted;
        if (fault) atomicOr(&a.error[env], fault);
    }
}

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

hipError_t rows_step(const StepArgs &a, bool word64, hipStream_t s) {
    if (a.B == 0) return hipSuccess;
    const unsigned block = 256;
    const uint64_t threads = a.B << a.log2L;
    size_t lds = (a.num_actions * sizeof(GateEntry) <= 32768) ? a.num_actions * sizeof(GateEntry) : 0;
    if (word64)
        hipLaunchKernelGGL(rows_step_kernel<uint64_t>, dim3(grid_for(threads, block)), dim3(block), lds, s, a);
    else
        hipLaunchKernelGGL(rows_step_kernel<uint32_t>, dim3(grid_for(threads, block)), dim3(block), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// init: constructor state / set_state / reset scramble
// ------------------------------------------------------------------------------------------
template <typename W>
__global__ __launch_bounds__(256) void rows_init_kernel(InitArgs a) {
    constexpr uint32_t RPL = RowTraits<W>::RPL;
    const uint32_t L = 1u << a.log2L;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env_raw = gid >> a.log2L;
    const uint32_t lie = (uint32_t)gid & (L - 1);
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    const uint32_t base = lane & ~(L - 1);
    const uint64_t env = env_raw < a.B ? env_raw : a.B - 1;
    // lanes of envs that are skipped (tail, or live episodes under qg_vec_reset_done) still take
    // part in the shuffles and the ballot below; they just store nothing
    const bool valid = env_raw < a.B && !(a.only_done && !a.done[env]);

    W ident[RPL], r[RPL];
    identity_rows<W>(ident, lie, a.D);
#pragma unroll
    for (int c = 0; c < (int)RPL; ++c) r[c] = ident[c];  // CFState::new (clifford.rs:35-42)

    if (a.mode == 1) {  // set_state (clifford.rs:299-304): data[i] = x > 0
#pragma unroll
        for (int c = 0; c < (int)RPL; ++c) {
            uint32_t row = lie * RPL + c;
            W w = 0;
            if (row < a.D) {
                if (a.format == QG_FMT_PACKED) {
                    w = reinterpret_cast<const W *>(a.src)[env * a.src_stride + row];
                    if (a.D < sizeof(W) * 8) w &= (((W)1 << a.D) - 1);
                } else if (a.format == QG_FMT_I64) {
                    const int64_t *p = reinterpret_cast<const int64_t *>(a.src) + env * a.src_stride + (uint64_t)row * a.D;
                    for (uint32_t col = 0; col < a.D; ++col) w |= (W)(p[col] > 0) << col;
                } else {
                    const int8_t *p = reinterpret_cast<const int8_t *>(a.src) + env * a.src_stride + (uint64_t)row * a.D;
                    for (uint32_t col = 0; col < a.D; ++col) w |= (W)(p[col] > 0) << col;
                }
            }
            r[c] = w;
        }
    } else if (a.mode == 2) {  // reset (clifford.rs:306-316): `difficulty` uniform draws, state only
        for (uint32_t t = 0; t < a.n_draws; ++t) {
            int64_t act = a.actions ? (int64_t)a.actions[(uint64_t)t * a.B + env]
                                    : (int64_t)rng_action(init_seed(a), a.env_base + env, t, a.num_actions);
            uint32_t ops = (act >= 0 && act < (int64_t)a.num_actions) ? a.gates[act].ops : 0u;
            apply_ops<W>(r, ops, lie, base);
        }
    }

    bool ok = true;
#pragma unroll
    for (int c = 0; c < (int)RPL; ++c) ok &= (r[c] == ident[c]);
    const uint64_t gmask = (L >= 64) ? ~0ull : (((1ull << L) - 1ull) << base);
    const bool solved = (__ballot(ok) & gmask) == gmask;

    if (valid) {
        store_rows<W>(reinterpret_cast<char *>(a.state) + ((env << a.log2L) + lie) * 16, r);
        if (lie == 0) {  // reset_internals (clifford.rs:272-283)
            a.depth[env] = a.depth_value;
            a.success[env] = (uint8_t)solved;
            a.reward[env] = solved ? 1.0f : 0.0f;
            a.done[env] = (uint8_t)(a.depth_value == 0 || solved);
            a.inverted[env] = 0;
            a.error[env] = 0;
            a.sol_len[env * 2] = 0;
            a.sol_len[env * 2 + 1] = 0;
            if (a.layers) {
                int32_t *lay = a.layers + env * a.layers_len;
                for (uint32_t i = 0; i + 2 < a.layers_len; ++i) lay[i] = -1;  // metrics.rs:47-52
                lay[a.layers_len - 2] = 0;
                lay[a.layers_len - 1] = 0;
            }
        }
    }
}

hipError_t rows_init(const InitArgs &a, bool word64, hipStream_t s) {
    if (a.B == 0) return hipSuccess;
    const unsigned block = 256;
    const uint64_t threads = a.B << a.log2L;
    if (word64)
        hipLaunchKernelGGL(rows_init_kernel<uint64_t>, dim3(grid_for(threads, block)), dim3(block), 0, s, a);
    else
        hipLaunchKernelGGL(rows_init_kernel<uint32_t>, dim3(grid_for(threads, block)), dim3(block), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// export: dense int8 observation / i64 entries / packed rows
// ------------------------------------------------------------------------------------------
// One thread per output row: reads one word, writes D contiguous bytes (or D int64, or 1 word).
template <typename W>
__global__ __launch_bounds__(256) void rows_export_kernel(ObsArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid / a.D;
    const uint32_t row = (uint32_t)(gid % a.D);
    if (env >= a.B) return;
    const uint32_t rows_per_env = (uint32_t)((16u / sizeof(W)) << a.log2L);
    const W w = reinterpret_cast<const W *>(a.state)[env * rows_per_env + row];
    if (a.format == QG_FMT_PACKED) {
        reinterpret_cast<W *>(a.out)[env * a.out_stride + row] = w;
    } else if (a.format == QG_FMT_I64) {
        int64_t *o = reinterpret_cast<int64_t *>(a.out) + env * a.out_stride + (uint64_t)row * a.D;
        for (uint32_t c = 0; c < a.D; ++c) o[c] = (int64_t)((w >> c) & 1);
    } else {
        int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride + (uint64_t)row * a.D;
        for (uint32_t c = 0; c < a.D; ++c) o[c] = (int8_t)((w >> c) & 1);
    }
}

// Dense int8 observation, D == 32 fast path (the CliffordEnv N=16 headline shape): one lane
// expands one packed row into 32 bytes = two 16 B stores; a wave writes 2 KiB contiguously.
__global__ __launch_bounds__(256) void rows_dense32_kernel(ObsArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per row
    if (gid >= a.B * 32ull) return;
    const uint32_t w = reinterpret_cast<const uint32_t *>(a.state)[gid];
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t nib = (w >> (4 * k)) & 0xFu;
        // spread 4 bits into 4 bytes: bit i -> byte i
        o[k] = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
    }
    uint4 *out = reinterpret_cast<uint4 *>(reinterpret_cast<int8_t *>(a.out) + gid * 32ull);
    out[0] = make_uint4(o[0], o[1], o[2], o[3]);
    out[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

hipError_t rows_export(const ObsArgs &a, bool word64, hipStream_t s) {
    if (a.B == 0) return hipSuccess;
    const unsigned block = 256;
    if (!word64 && a.D == 32 && a.format == QG_FMT_U8 && a.log2L == 3 && a.out_stride == 1024 &&
        (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
        hipLaunchKernelGGL(rows_dense32_kernel, dim3(grid_for(a.B * 32ull, block)), dim3(block), 0, s, a);
        return hipGetLastError();
    }
    const uint64_t threads = a.B * a.D;
    if (word64)
        hipLaunchKernelGGL(rows_export_kernel<uint64_t>, dim3(grid_for(threads, block)), dim3(block), 0, s, a);
    else
        hipLaunchKernelGGL(rows_export_kernel<uint32_t>, dim3(grid_for(threads, block)), dim3(block), 0, s, a);
    return hipGetLastError();
}

// Env::masks (clifford.rs:349-351)
__global__ __launch_bounds__(256) void masks_kernel(const uint8_t *success, uint8_t *out, uint64_t total, uint32_t A) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < total) out[gid] = success[gid / A] ? 0 : 1;
}
// Indices of the finished envs, packed (qg_vec_reset_done): a wave ballots its `done` flags, one lane
// reserves that many slots of `list` with an atomic add, every done lane writes its env index at its
// prefix.  The order across waves is arbitrary -- every env's reset depends on (seed, env) only.
__global__ __launch_bounds__(256) void compact_done_kernel(const uint8_t *done, uint64_t B, uint32_t *list, uint32_t *count) {
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    const bool d = env < B && done[env];
    const uint64_t m = __ballot(d);
    if (!m) return;
    uint32_t base = 0;
    if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (d) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)env;
}
// `count` must be zero on entry: the last kernel that consumes the list zeroes it again (list_count_take)
hipError_t compact_done(const uint8_t *done, uint64_t B, uint32_t *list, uint32_t *count, hipStream_t s) {
    hipLaunchKernelGGL(compact_done_kernel, dim3(grid_for(B, 256)), dim3(256), 0, s, done, B, list, count);
    return hipGetLastError();
}

hipError_t masks_fill(const uint8_t *success, uint8_t *out, uint64_t B, uint32_t A, hipStream_t s) {
    const uint64_t total = B * A;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(masks_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, success, out, total, A);
    return hipGetLastError();
}

}  // namespace qg
