// kernels_perm.hip -- PermutationEnv with more than 16 qubits (PERMB layout: one byte per entry, N <= 256).
//
// Reference semantics: rust/src/envs/permutation.rs -- `state: Vec<usize>` of any length (:29-60), SWAP(q1, q2) swaps
// state[q1] and state[q2] (:110-114, other gate kinds leave the state alone :205-208), solved = state[i] == i for all i
// (:122-128), invert_perm: inv[perm[i]] = i (:101-107), step order: gate, solution push only for a valid action (:210-216),
// maybe_random_invert BEFORE the depth decrement (:219-221), set_state casts `x as usize` (:168-173), observe = indices
// i * N + state[i] (:241-243).  The nibble-packed one-word kernel (kernels_small.hip) keeps N <= 16.
//
// Memory (PERMB layout): envs in tiles of 64 (one wavefront), a tile = NG groups of 1 KiB, group g holds for lane l the 16 bytes
// state[16 g .. 16 g + 15] of env tile * 64 + l (NG = ceil(N / 16); bytes past N hold their own index, so the padding is a fixed
// point of everything).  A wave's group load / store is one contiguous 1 KiB.
//
// Mapping: thread per env.
//   * one step per launch without add_inverts (`permb_step1_kernel`): a SWAP is two byte gathers and two byte scatters at per-lane
//     addresses; `solved` comes from a per-env count of non-fixed points kept incrementally (a swap changes two entries; a
//     permutation and its inverse have the same number of fixed points, so inversion leaves the count alone);
//   * add_inverts and fused rollouts (`permb_step_kernel`): the env's bytes live in LDS as dwords [d][lane] (bank = lane: any
//     per-lane index is conflict-free), a SWAP is two byte reads and two byte writes, inversion scatters i to a second LDS
//     image at index state[i] and the two images trade places.
// HBM-bound: N R + <= 2 W bytes (one step) / 2 x roundup(N, 16) bytes (inversion) + 16 B of scalars per env-step.
#include "device_common.hpp"
#include "qgym_plan.hpp"

namespace qg {

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

// a non-SWAP gate, an out-of-range action or SWAP(q, q) leaves the state alone; `desc` = kind | q0 << 8 | q1 << 16
__device__ inline bool permb_is_swap(uint32_t desc, uint32_t &q0, uint32_t &q1) {
    q0 = (desc >> 8) & 0xFFu;
    q1 = (desc >> 16) & 0xFFu;
    return (desc & 0xFFu) == QG_SWAP && q0 != q1;
}

__device__ inline uint8_t *permb_byte(void *state, uint64_t env, uint32_t ng, uint32_t i) {
    return reinterpret_cast<uint8_t *>(state) + ((env >> 6) * ng + (i >> 4)) * 1024u + (env & 63u) * 16u + (i & 15u);
}

// ---- one step per launch, no inversion ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void permb_step1_kernel(StepArgs a, uint32_t ng) {
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    QG_PREFETCH_STEP_ARGS(a);
    if (env >= a.B) return;
    const int64_t act = load_action(a.actions, env, a.flags & F_ACT64);
    int32_t depth = a.depth[env];
    const uint32_t nbad0 = a.bad[env];
    uint32_t nbad = nbad0, fault = 0;
    const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // gateset.get(action) (permutation.rs:197)
    float penalty = 0.0f;
    if (in_range) {
        penalty = a.gates[act].penalty;
        const uint32_t desc = a.descs[act];
        if (a.flags & F_LAYERS) penalty = layers_penalty(layer_rec(a.layers, env, 2 * a.N + 2), a.N, desc, a.w);
        uint32_t q0, q1;
        if (permb_is_swap(desc, q0, q1)) {  // permutation.rs:110-114
            uint8_t *p0 = permb_byte(a.state, env, ng, q0), *p1 = permb_byte(a.state, env, ng, q1);
            const uint32_t v0 = *p0, v1 = *p1;
            *p0 = (uint8_t)v1;
            *p1 = (uint8_t)v0;
            nbad = nbad - (uint32_t)(v0 != q0) - (uint32_t)(v1 != q1) + (uint32_t)(v1 != q0) + (uint32_t)(v0 != q1);
        }
        if (a.flags & F_TRACK) {  // pushed only for a valid action (permutation.rs:210-216)
            const int32_t n = a.sol_len[env * 2];
            if ((uint32_t)n < a.sol_cap) {
                sol_at(a, env, (uint32_t)n) = sol_word_framed(act, false);
                a.sol_len[env * 2] = n + 1;
            } else {
                fault |= 8u;
            }
        }
    }
    depth = depth > 0 ? depth - 1 : 0;  // permutation.rs:221
    const bool solved = nbad == 0;      // :222
    const float achieved = solved ? 1.0f : 0.0f;
    const float reward = achieved - penalty;  // :223-224
    if (a.rewards_seq) a.rewards_seq[env] = reward;
    if (a.dones_seq) a.dones_seq[env] = (uint8_t)(depth == 0 || solved);
    if (nbad != nbad0) a.bad[env] = nbad;
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (fault) atomicOr(&a.error[env], fault);
}

// the env's bytes in LDS: dword d of lane l at img[d * lanes + l]
__device__ inline uint32_t lds_get_byte(const uint32_t *img, uint32_t lanes, uint32_t l, uint32_t i) {
    return (img[(i >> 2) * lanes + l] >> (8u * (i & 3u))) & 0xFFu;
}
__device__ inline void lds_put_byte(uint32_t *img, uint32_t lanes, uint32_t l, uint32_t i, uint32_t v) {
    reinterpret_cast<uint8_t *>(img + (i >> 2) * lanes + l)[i & 3u] = (uint8_t)v;
}

__device__ inline void permb_load_image(const void *state, uint64_t env, uint32_t ng, uint32_t *img, uint32_t lanes, uint32_t l) {
    const uint4 *tile = reinterpret_cast<const uint4 *>(state) + (env >> 6) * (uint64_t)ng * 64u + (env & 63u);
    for (uint32_t g = 0; g < ng; ++g) {
        const uint4 v = tile[g * 64u];
        img[(4 * g + 0) * lanes + l] = v.x;
        img[(4 * g + 1) * lanes + l] = v.y;
        img[(4 * g + 2) * lanes + l] = v.z;
        img[(4 * g + 3) * lanes + l] = v.w;
    }
}
__device__ inline void permb_store_image(void *state, uint64_t env, uint32_t ng, const uint32_t *img, uint32_t lanes, uint32_t l) {
    uint4 *tile = reinterpret_cast<uint4 *>(state) + (env >> 6) * (uint64_t)ng * 64u + (env & 63u);
    for (uint32_t g = 0; g < ng; ++g)
        tile[g * 64u] = make_uint4(img[(4 * g + 0) * lanes + l], img[(4 * g + 1) * lanes + l], img[(4 * g + 2) * lanes + l], img[(4 * g + 3) * lanes + l]);
}
__device__ inline uint32_t permb_count_bad(const uint32_t *img, uint32_t lanes, uint32_t l, uint32_t N) {
    uint32_t nbad = 0;
    for (uint32_t i = 0; i < N; ++i) nbad += (uint32_t)(lds_get_byte(img, lanes, l, i) != i);
    return nbad;
}

// ---- add_inverts and / or T steps per launch: state resident in LDS --------------------------------------------------------------
__global__ __launch_bounds__(256) void permb_step_kernel(StepArgs a, uint32_t ng) {
    extern __shared__ uint32_t permb_lds[];  // two images of 4 * ng dwords per lane
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lanes = blockDim.x, l = threadIdx.x;
    if (env >= a.B) return;  // no barrier below: a lane only ever touches its own LDS column
    uint32_t *cur = permb_lds, *alt = permb_lds + 4u * ng * lanes;
    const bool act64 = a.flags & F_ACT64;
    permb_load_image(a.state, env, ng, cur, lanes, l);
    int32_t depth = a.depth[env];
    uint32_t inverted = (a.flags & F_INVERTS) ? a.inverted[env] : 0u;
    uint32_t nbad = a.bad ? a.bad[env] : permb_count_bad(cur, lanes, l, a.N);
    int32_t nf = (a.flags & F_TRACK) ? a.sol_len[env * 2] : 0, nb = (a.flags & F_TRACK) ? a.sol_len[env * 2 + 1] : 0;
    bool solved = false, dirty = false;
    float reward = 0.0f;
    uint32_t fault = 0;
    for (uint32_t t = 0; t < a.T; ++t) {
        const int64_t act = load_action(a.actions, (uint64_t)t * a.B + env, act64);
        const bool in_range = act >= 0 && act < (int64_t)a.num_actions;
        float penalty = 0.0f;
        if (in_range) {
            penalty = a.gates[act].penalty;
            const uint32_t desc = a.descs[act];
            if (a.flags & F_LAYERS) penalty = layers_penalty(layer_rec(a.layers, env, 2 * a.N + 2), a.N, desc, a.w);
            uint32_t q0, q1;
            if (permb_is_swap(desc, q0, q1)) {
                const uint32_t v0 = lds_get_byte(cur, lanes, l, q0), v1 = lds_get_byte(cur, lanes, l, q1);
                lds_put_byte(cur, lanes, l, q0, v1);
                lds_put_byte(cur, lanes, l, q1, v0);
                nbad = nbad - (uint32_t)(v0 != q0) - (uint32_t)(v1 != q1) + (uint32_t)(v1 != q0) + (uint32_t)(v0 != q1);
                dirty = true;
            }
            if (a.flags & F_TRACK) {  // permutation.rs:210-216: into solution_inv while inverted
                if ((uint32_t)(nf + nb) < a.sol_cap) {
                    sol_at(a, env, (uint32_t)(nf + nb)) = sol_word_framed(act, inverted);
                    if (inverted) ++nb;
                    else ++nf;
                } else {
                    fault |= 8u;
                }
            }
        }
        if (a.flags & F_INVERTS) {  // maybe_random_invert (permutation.rs:186-192), before the depth decrement (:219-221)
            const uint32_t coin = a.coins ? a.coins[(uint64_t)t * a.B + env]
                                          : (uint32_t)(rng_draw(a.seed ^ 0x636F696Eull, a.env_base + env, step_clock(a) + t) >> 63);
            if (coin & 1u) {
                const uint32_t nb16 = 16u * ng;
                // `let mut inv = vec![0; n]; inv[perm[i]] = i` (permutation.rs:101-107); padding bytes map to themselves
                for (uint32_t d = 0; d < 4u * ng; ++d) {
                    const uint32_t idw = (4u * d) | ((4u * d + 1u) << 8) | ((4u * d + 2u) << 16) | ((4u * d + 3u) << 24);
                    uint32_t keep = 0;  // byte mask of the entries >= N
                    for (uint32_t k = 0; k < 4; ++k) keep |= (4u * d + k >= a.N) ? 0xFFu << (8u * k) : 0u;
                    alt[d * lanes + l] = idw & keep;
                }
                for (uint32_t i = 0; i < a.N; ++i) {
                    const uint32_t p = lds_get_byte(cur, lanes, l, i);
                    if (p < nb16) lds_put_byte(alt, lanes, l, p, i);
                }
                uint32_t *tmp = cur;
                cur = alt;
                alt = tmp;
                inverted ^= 1u;
                dirty = true;  // the number of non-fixed points is that of the inverse
            }
        }
        depth = depth > 0 ? depth - 1 : 0;
        solved = nbad == 0;
        const float achieved = solved ? 1.0f : 0.0f;
        reward = achieved - penalty;
        if (a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
        if (a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
    }
    if (dirty) permb_store_image(a.state, env, ng, cur, lanes, l);
    if (a.bad) a.bad[env] = nbad;
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (a.flags & F_INVERTS) a.inverted[env] = (uint8_t)inverted;
    if (a.flags & F_TRACK) {
        a.sol_len[env * 2] = nf;
        a.sol_len[env * 2 + 1] = nb;
    }
    if (fault) atomicOr(&a.error[env], fault);
}

// ---- constructor state / set_state / reset / reset_done ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void permb_init_kernel(InitArgs a, uint32_t ng, const uint32_t *descs) {
    extern __shared__ uint32_t permb_lds[];
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lanes = blockDim.x, l = threadIdx.x;
    if (env >= a.B) return;
    if (a.only_done && !a.done[env]) return;  // qg_vec_reset_done
    uint32_t *img = permb_lds;
    for (uint32_t d = 0; d < 4u * ng; ++d) img[d * lanes + l] = (4u * d) | ((4u * d + 1u) << 8) | ((4u * d + 2u) << 16) | ((4u * d + 3u) << 24);  // identity (permutation.rs:77)
    uint32_t fault = 0;
    if (a.mode == 1) {  // set_state (permutation.rs:168-173): state[i] = x as usize
        for (uint32_t i = 0; i < a.N; ++i) {
            int64_t v;
            if (a.format == QG_FMT_I64) v = reinterpret_cast<const int64_t *>(a.src)[env * a.src_stride + i];
            else v = reinterpret_cast<const uint8_t *>(a.src)[env * a.src_stride + i];  // U8 / PACKED: one byte per entry
            if (v < 0 || v >= (int64_t)a.N) {  // the reference would index out of bounds at the next observe / invert
                fault |= QG_FAULT_BAD_STATE;
                v = 0;
            }
            lds_put_byte(img, lanes, l, i, (uint32_t)v);
        }
    } else if (a.mode == 2) {  // reset: `difficulty` random gates on the identity (permutation.rs:175-184)
        const uint64_t seed = init_seed(a);
        for (uint32_t t = 0; t < a.n_draws; ++t) {
            const int64_t act = a.actions ? (int64_t)a.actions[(uint64_t)t * a.B + env] : (int64_t)rng_action(seed, a.env_base + env, t, a.num_actions);
            if (act < 0 || act >= (int64_t)a.num_actions) continue;
            uint32_t q0, q1;
            if (permb_is_swap(descs[act], q0, q1)) {
                const uint32_t v0 = lds_get_byte(img, lanes, l, q0), v1 = lds_get_byte(img, lanes, l, q1);
                lds_put_byte(img, lanes, l, q0, v1);
                lds_put_byte(img, lanes, l, q1, v0);
            }
        }
    }
    // permutation.rs:168-173 stores the vector unvalidated, and step / observe / solved work on any vector of in-range entries (the count
    // of non-fixed points is kept per swapped position).  Only inversion needs a permutation -- a repeated entry has no inverse and the
    // count is no longer that of the inverse -- so such a state is a fault only with add_inverts (same rule in kernels_small.hip)
    if (a.mode == 1 && a.inverts) {
        for (uint32_t i = 0; i < a.N && !(fault & QG_FAULT_BAD_STATE); ++i) {
            const uint32_t v = lds_get_byte(img, lanes, l, i);
            for (uint32_t j = i + 1; j < a.N; ++j)
                if (lds_get_byte(img, lanes, l, j) == v) fault |= QG_FAULT_BAD_STATE;
        }
    }
    const uint32_t nbad = permb_count_bad(img, lanes, l, a.N);
    const bool solved = nbad == 0;
    permb_store_image(a.state, env, ng, img, lanes, l);
    if (a.bad) a.bad[env] = nbad;
    a.depth[env] = a.depth_value;  // reset_internals (permutation.rs:134-145)
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

// ---- observe / get_state: one thread per (env, entry) ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void permb_export_kernel(ObsArgs a, uint32_t ng) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid / a.N;
    if (env >= a.B) return;
    const uint32_t i = (uint32_t)(gid - env * a.N);
    const uint32_t v = *permb_byte(const_cast<void *>(a.state), env, ng, i);
    if (a.format == QG_FMT_I64) {  // get_state (permutation.rs:130-132)
        reinterpret_cast<int64_t *>(a.out)[env * a.out_stride + i] = (int64_t)v;
    } else if (a.format == QG_FMT_PACKED) {
        reinterpret_cast<uint8_t *>(a.out)[env * a.out_stride + i] = (uint8_t)v;
    } else {  // observe: indices i * N + state[i] (permutation.rs:241-243), densified row by row
        int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride + (uint64_t)i * a.N;
        for (uint32_t c = 0; c < a.N; ++c) o[c] = (int8_t)(c == v);
    }
}

static unsigned permb_block(uint32_t ng, uint32_t images) {  // threads per block so that `images` LDS images fit 32 KiB
    unsigned t = 256;
    while (t > 64 && (size_t)images * 16u * ng * t > 32768u) t >>= 1;
    return t;
}

hipError_t permb_step(const StepArgs &a, const uint32_t ng, hipStream_t s) {
    if (!a.B) return hipSuccess;
    if (plan::permb_step_kernel_of(a.flags, a.T, a.bad != nullptr) == plan::SK_PERMB_STEP1) {  // qgym_plan.hpp
        hipLaunchKernelGGL(permb_step1_kernel, dim3(grid_for(a.B, 256)), dim3(256), 0, s, a, ng);
        return hipGetLastError();
    }
    const unsigned block = permb_block(ng, 2);
    hipLaunchKernelGGL(permb_step_kernel, dim3(grid_for(a.B, block)), dim3(block), (size_t)2 * 16u * ng * block, s, a, ng);
    return hipGetLastError();
}
hipError_t permb_init(const InitArgs &a, uint32_t ng, const uint32_t *descs, hipStream_t s) {
    if (!a.B) return hipSuccess;
    const unsigned block = permb_block(ng, 1);
    hipLaunchKernelGGL(permb_init_kernel, dim3(grid_for(a.B, block)), dim3(block), (size_t)16u * ng * block, s, a, ng, descs);
    return hipGetLastError();
}
hipError_t permb_export(const ObsArgs &a, uint32_t ng, hipStream_t s) {
    if (!a.B) return hipSuccess;
    hipLaunchKernelGGL(permb_export_kernel, dim3(grid_for(a.B * a.N, 256)), dim3(256), 0, s, a, ng);
    return hipGetLastError();
}

}  // namespace qg
