// kernels_policy.hip -- the policy's first layer computed straight from the resident bit-packed state
// ("policy forward consuming packed obs directly").
//
// The reference's policy (twisterl.nn.BasicPolicy as configured by rl/configs.py:531-607; checkpoint
// shapes in examples/models/*.pt) starts with Linear(prod(obs_shape) -> 512) on the flattened dense
// {0,1} observation.  A collector that materialises that observation in bf16 writes and re-reads
// 2 KiB per env per step for 128 B of information (CliffordEnv 16q), and the GEMM behind it streams
// both operands through LDS.  Here the A operand never exists in memory:
//
//   h[e][n] = act( sum_k obs[e][k] * W[n][k] + b[n] ),   obs[e][row * D + col] = bit `col` of row `row`
//
//   * obs bits come from the TILE layout (kernels_qm.hip) 16 bytes per lane at a time and become MFMA A
//     fragments in registers.  A bf16 with a single exponent bit set is a power of two (bit 14: 2, bit 13:
//     2^-63, bit 12: 2^-95, bit 11: 2^-111), so the four VGPRs of a fragment are `rotr(word, t) & mask_j`
//     with mask_j = 0x40004000 >> j: five VALU ops per fragment, no table, no LDS.  The weights are packed
//     once into the k order those rotations yield, element j pre-multiplied by 2^0 / 2^64 / 2^96 / 2^112
//     (exact in bf16), so every product is exactly 2 W and the factor 2 leaves in the epilogue.
//   * a workgroup owns ONE 64-column slab of W for its whole life: [64][32 * R] bf16 <= 128 KiB, loaded
//     into LDS once (global_load_lds), already in fragment order (lane-linear ds_read_b128, conflict-free).
//     The main loop has no barrier and no global->LDS traffic; block b takes slab b % n_slabs, so the
//     blocks that share an XCD's L2 share a slab.
//   * a wave computes 64 envs x 64 columns per pass: per k-step of 16, two B fragments from LDS, two A
//     fragments from registers, four v_mfma_f32_32x32x16_bf16; two waves per SIMD.  What limits the loop is
//     the SIMD's issue port (an MFMA holds it for 8 of its 32 cycles, a VALU op for 4), so everything
//     beside the MFMAs is kept to ~4 VALU ops per MFMA: the expansion above, pass-invariant addressing with
//     immediate offsets, the first MFMA of a pass taking C = 0 instead of a cleared accumulator.
//   * epilogue: x 0.5 + bias (one fma), optional ReLU, packed bf16 conversion, then a 4x4 transpose over
//     the lane quad (DPP) so that a lane holds 8 adjacent outputs of one row: 16-byte stores, 128 contiguous
//     bytes per row.  The stores are deferred and leave one at a time under the next pass's groups -- a burst
//     of stores at every pass end (all workgroups reach it together) stalls the bit loads queued behind it.
//
// Numerics: products are exact, accumulation is the MFMA's f32 chain; the test compares against an f64
// reference within bf16 output rounding and pins the k permutation with integer data.  |W| >= 2^16 saturates
// in the packed form (2^112 W must stay finite in bf16).
#include <hip/hip_bf16.h>

#include <cstring>

#include "device_common.hpp"
#include "qgym_host.hpp"
#include "qm_step1.hpp"

namespace qg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr uint32_t EMB_SLAB = 64;       // output columns per workgroup
constexpr uint32_t EMB_MA = 2;          // 32-env MFMA row tiles per wave
constexpr uint32_t EMB_WAVES = 8;       // waves per workgroup (two per SIMD)
constexpr uint32_t EMB_WAVE_ENVS = 32 * EMB_MA;
constexpr uint32_t EMB_BLOCK_ENVS = EMB_WAVES * EMB_WAVE_ENVS;
constexpr uint32_t EMB_THREADS = 64 * EMB_WAVES;

// matrix row of TILE slot `slot` (kernels_qm.hip: X-type row j = slot 2j, Z-type row N+j = slot 2j+1), or -1
__host__ __device__ inline int32_t emb_slot_row(uint32_t slot, uint32_t N, bool has_z) {
    if (!has_z) return slot < N ? (int32_t)slot : -1;
    const uint32_t j = slot >> 1;
    if (j >= N) return -1;
    return (int32_t)((slot & 1u) ? N + j : j);
}

// k-steps of a packed slab: 8 per 16-byte group of the env, groups padded to an even count (the kernel's
// bit buffers alternate between two register sets)
__host__ __device__ inline uint32_t emb_groups(uint32_t R) { return ((R / 4u) + 1u) & ~1u; }
__host__ __device__ inline uint32_t emb_ksteps(uint32_t R) { return 8u * emb_groups(R); }

// Packed weights: [slab][k-step s < emb_ksteps(R)][nb < 2][lane < 64][e < 8] bf16.  Lane (c = lane & 31, h = lane >> 5)
// element e = 2j + half of k-step s multiplies bit ((half ? 30 : 14) - j + 8 (s & 1) + 4h) mod 32 of slot s >> 1,
// carries the factor 2^(128 - (128 >> j)) (the kernel's A element for it is 2^((128 >> j) - 127)) and belongs to
// output column slab * 64 + 2c + nb.
template <typename WT>
__global__ __launch_bounds__(256) void pack_embed_kernel(const WT *w, uint64_t ld, uint32_t hidden, uint32_t R, uint32_t N, uint32_t D,
                                                         uint32_t has_z, __hip_bfloat16 *out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t KS = emb_ksteps(R);
    const uint64_t total = (uint64_t)(hidden / EMB_SLAB) * KS * 2 * 64 * 8;
    if (idx >= total) return;
    const uint32_t e = (uint32_t)idx & 7u, lane = (uint32_t)(idx >> 3) & 63u, nb = (uint32_t)(idx >> 9) & 1u;
    const uint64_t ss = idx >> 10;
    const uint32_t s = (uint32_t)(ss % KS), slab = (uint32_t)(ss / KS);
    const uint32_t c = lane & 31u, h = lane >> 5, j = e >> 1, half = e & 1u;
    const uint32_t pos = ((half ? 30u : 14u) - j + 8u * (s & 1u) + 4u * h) & 31u;
    const int32_t row = (s >> 1) < R ? emb_slot_row(s >> 1, N, has_z != 0) : -1;  // k-steps past 2R pad the slab to an even group count
    const uint32_t n = slab * EMB_SLAB + 2u * c + nb;
    float v = 0.0f;
    if (row >= 0 && pos < D) {
        v = (float)w[(uint64_t)n * ld + (uint64_t)row * D + pos];
        v *= __uint_as_float((127u + 128u - (128u >> j)) << 23);  // 2^(128 - (128 >> j)): 1, 2^64, 2^96, 2^112
        const float lim = 3.3895313892515355e38f;                   // largest finite bf16
        v = v > lim ? lim : v < -lim ? -lim : v;
    }
    out[idx] = __float2bfloat16(v);
}

struct EmbedArgs {
    const uint4 *state;     // TILE layout
    const uint4 *wp;        // packed weights
    const float *bias;      // [hidden] f32 or null
    uint32_t *out;          // [B][ld_out / 2] bf16 pairs
    uint4 *dump;            // 64 x 16 B that nobody reads: where the lanes of rows past B (and the "nothing pending yet" case) store
    uint64_t B;
    uint64_t ld_out;        // elements per env row of out
    uint32_t n_slabs;
    uint32_t relu;
    uint32_t *obs;          // embed_small_kernel: also the packed observation [B][D] (what qg_vec_observe_packed writes), or null
    uint32_t N, D, has_z;
    // LFD layout (LinearFunctionEnv with add_inverts, kernels_lfd.hip): a tile holds two regions of G groups, bit 0 of the env's
    // `inverted` byte says which one is the state.  TILE layout: region == null, tile_groups == G.
    const uint8_t *region;
    uint32_t tile_groups;   // 16-byte groups per env in a tile
};

// A fragment (8 bf16 for this lane's row and k-half) from a row word: see the header comment
__device__ __forceinline__ bf16x8 emb_expand(uint32_t w, uint32_t sh) {
    const uint32_t t = __builtin_amdgcn_alignbit(w, w, sh);
    u32x4 v;
    v.x = t & 0x40004000u;
    v.y = t & 0x20002000u;
    v.z = t & 0x10001000u;
    v.w = t & 0x08000800u;
    return __builtin_bit_cast(bf16x8, v);
}

// 2x2 transpose step of the epilogue: lanes l and l ^ m (m = 1: DPP quad_perm [1,0,3,2] = 0xB1; m = 2: [2,3,0,1] = 0x4E)
// exchange so that the lane with bit m clear ends with {a, partner's a} and the other with {partner's b, b}
template <int CTRL>
__device__ __forceinline__ void emb_quad_swap(uint32_t &a, uint32_t &b, bool upper) {
    const uint32_t x = upper ? a : b;
    const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
    a = upper ? y : a;
    b = upper ? b : y;
}

struct EmbWave {
    f32x16 acc[EMB_MA][2];
    bf16x8 a_c[EMB_MA], b_c[2];
};

// One group of a pass (8 k-steps = one 16-byte group of every env of the wave's row tiles).  `cur` holds the
// group's bits, `nxt` receives the following group's (of this pass or of the next one): its loads are issued
// first and consumed only by the last phase, which expands the first word of `nxt`.  `after_loads` issues the
// deferred output stores of the previous pass that belong to this group.  Software pipeline: while the MFMAs
// of k-step s issue, the B fragments of s + 1 are read from LDS and its A fragments are expanded into a second
// register set.  FIRST: the pass starts here, the accumulators start from C = 0.
template <bool FIRST, typename F>
__device__ __forceinline__ void emb_group(EmbWave &w, const uint4 (&cur)[EMB_MA], uint4 (&nxt)[EMB_MA], const uint4 *(&pn)[EMB_MA],
                                          uint32_t pn_off, const uint4 *bl, uint32_t wrap, const uint32_t (&sh)[2], F &&after_loads) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (uint32_t i = 0; i < EMB_MA; ++i) nxt[i] = pn[i][pn_off];
    after_loads();
#pragma unroll
    for (uint32_t ss = 0; ss < 8; ++ss) {
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t sn = ss + 1, compn = (sn >> 1) & 3u, spn = sn & 1u;
        bf16x8 a_n[EMB_MA], b_n[2];
        if (ss < 7) {
            b_n[0] = __builtin_bit_cast(bf16x8, bl[(sn * 2u + 0u) * 64u]);
            b_n[1] = __builtin_bit_cast(bf16x8, bl[(sn * 2u + 1u) * 64u]);
#pragma unroll
            for (uint32_t i = 0; i < EMB_MA; ++i) {
                const uint32_t word = compn == 0 ? cur[i].x : compn == 1 ? cur[i].y : compn == 2 ? cur[i].z : cur[i].w;
                a_n[i] = emb_expand(word, sh[spn]);
            }
        } else {  // the next group's first k-step (k-step 0 again after the last group)
            b_n[0] = __builtin_bit_cast(bf16x8, (bl - wrap)[(8u * 2u + 0u) * 64u]);
            b_n[1] = __builtin_bit_cast(bf16x8, (bl - wrap)[(8u * 2u + 1u) * 64u]);
#pragma unroll
            for (uint32_t i = 0; i < EMB_MA; ++i) a_n[i] = emb_expand(nxt[i].x, sh[0]);
        }
#pragma unroll
        for (uint32_t i = 0; i < EMB_MA; ++i) {
            if (FIRST && ss == 0) {
                const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                w.acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a_c[i], w.b_c[0], zero, 0, 0, 0);
                w.acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a_c[i], w.b_c[1], zero, 0, 0, 0);
            } else {
                w.acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a_c[i], w.b_c[0], w.acc[i][0], 0, 0, 0);
                w.acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.a_c[i], w.b_c[1], w.acc[i][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the two LDS reads first
#pragma unroll
        for (uint32_t m = 0; m < 2 * EMB_MA; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // a share of the next k-step's expansion
        }
#pragma unroll
        for (uint32_t i = 0; i < EMB_MA; ++i) w.a_c[i] = a_n[i];
        w.b_c[0] = b_n[0];
        w.b_c[1] = b_n[1];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// G = 16-byte groups per env (R / 4); the slab holds GP = G rounded up to even groups (the padding group has
// zero weights and re-reads group G - 1's bits).  The pass loop is unrolled over the groups, so bit-load offsets
// are immediates and which deferred store goes with which group is a compile-time fact.
template <uint32_t G>
__global__ __launch_bounds__(EMB_THREADS, 1) void embed_bits_kernel(EmbedArgs a) {
    extern __shared__ uint4 emb_lds[];  // the slab: [8 GP k-steps][2][64 lanes] x 16 B
    constexpr uint32_t GP = (G + 1u) & ~1u;
    constexpr uint32_t NQ = EMB_MA * 4u;  // output quads (16 B per lane) per pass
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t slab = blockIdx.x % a.n_slabs, mgroup = blockIdx.x / a.n_slabs, n_mgroups = gridDim.x / a.n_slabs;
    constexpr uint32_t group_vec = 8u * 2u * 64u;      // uint4 per group of 8 k-steps
    constexpr uint32_t slab_vec = GP * group_vec;      // uint4 per slab
    {   // the slab goes global -> LDS without passing through registers: 1 KiB per wave instruction, lane-linear
        const uint4 *src = a.wp + (uint64_t)slab * slab_vec;
        for (uint32_t c = wave * 64u; c < slab_vec; c += EMB_THREADS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c + lane),
                                             (__attribute__((address_space(3))) void *)(emb_lds + c), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t sh[2] = {4u * h, 8u + 4u * h};
    const uint32_t n0 = slab * EMB_SLAB + 2u * r;
    const float bias0 = a.bias ? a.bias[n0] : 0.0f, bias1 = a.bias ? a.bias[n0 + 1] : 0.0f;
    const uint64_t pass_stride = (uint64_t)n_mgroups * EMB_BLOCK_ENVS;

    uint64_t env0 = (uint64_t)mgroup * EMB_BLOCK_ENVS + (uint64_t)wave * EMB_WAVE_ENVS;  // the pass being computed
    if (env0 >= a.B) return;  // wave-uniform; no barrier below
    // group 0 of a pass for row tile i, as this lane reads it; later groups are immediate offsets from it
    auto point = [&](uint64_t e0, const uint4 *(&p)[EMB_MA]) {
#pragma unroll
        for (uint32_t i = 0; i < EMB_MA; ++i) {
            uint64_t env = e0 + 32u * i + r;
            env = env < a.B ? env : a.B - 1;  // tail: duplicate the last env, its rows are not stored
            p[i] = a.state + (env >> 6) * (uint64_t)(a.tile_groups * 64u) + (env & 63u) + (a.region ? (a.region[env] & 1u) * (G * 64u) : 0u);
        }
    };
    const uint4 *pc[EMB_MA], *pnx[EMB_MA];  // this pass, next pass (past the last pass: this pass again, the data is not used)
    point(env0, pc);
    EmbWave w;
    uint4 bufa[EMB_MA], bufb[EMB_MA];
#pragma unroll
    for (uint32_t i = 0; i < EMB_MA; ++i) bufa[i] = pc[i][0];
#pragma unroll
    for (uint32_t i = 0; i < EMB_MA; ++i) w.a_c[i] = emb_expand(bufa[i].x, sh[0]);
    w.b_c[0] = __builtin_bit_cast(bf16x8, emb_lds[lane]);
    w.b_c[1] = __builtin_bit_cast(bf16x8, emb_lds[64u + lane]);

    // outputs of the finished pass wait here (see the header comment)
    uint4 pend[NQ];
    uint32_t pend_rows = 0;  // rows of the pending pass that exist (0: nothing pending)
    const uint64_t ldw = a.ld_out >> 1;  // dwords per output row
    const uint32_t lane_row = (lane & 3u) + 4u * h;
    uint32_t *const out_lane = a.out + ((slab * EMB_SLAB + 8u * (r >> 2)) >> 1) + (uint64_t)lane_row * ldw;
    uint32_t *pend_ptr = out_lane;
    // Every store is unconditional (rows that do not exist go to the dump): behind a branch the compiler could not
    // count it and would make each wait for a group's bit loads also wait for the store issued after them.
    uint4 *const dump = a.dump + lane;
    auto store_quad = [&](uint32_t qi) {  // quad qi = (row tile i, register group j): row 32i + 8j + (lane & 3) + 4h of the pass
        const uint32_t row0 = 32u * (qi >> 2) + 8u * (qi & 3u);
        uint4 *dst = reinterpret_cast<uint4 *>(pend_ptr + (uint64_t)row0 * ldw);
        dst = lane_row + row0 < pend_rows ? dst : dump;
        *dst = pend[qi];
    };

    for (;;) {  // one pass per trip
        point(env0 + pass_stride < a.B ? env0 + pass_stride : env0, pnx);
#pragma unroll
        for (uint32_t g = 0; g < GP; ++g) {
            auto stores = [&]() {
#pragma unroll
                for (uint32_t qi = 0; qi < NQ; ++qi)
                    if (qi * GP / NQ == g) store_quad(qi);
            };
            constexpr uint32_t none = 0u;
            const uint32_t gn = g + 1u < GP ? (g + 1u < G ? g + 1u : G - 1u) : 0u;  // the group fetched under this one
            const uint4 *pn[EMB_MA];
#pragma unroll
            for (uint32_t i = 0; i < EMB_MA; ++i) pn[i] = g + 1u < GP ? pc[i] : pnx[i];
            if (g & 1u) emb_group<false>(w, bufb, bufa, pn, gn * 64u, emb_lds + g * group_vec + lane, g + 1u < GP ? none : slab_vec, sh, stores);
            else if (g == 0) emb_group<true>(w, bufa, bufb, pn, gn * 64u, emb_lds + g * group_vec + lane, none, sh, stores);
            else emb_group<false>(w, bufa, bufb, pn, gn * 64u, emb_lds + g * group_vec + lane, none, sh, stores);
        }
        // The pass is complete: x 0.5 + bias, ReLU, bf16.  C layout: column = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5),
        // so a lane holds one column pair (one dword) of 16 rows.  A 4x4 transpose over the lane quad (two DPP
        // stages) turns four dwords = four rows x one column pair into one row x four column pairs.
#pragma unroll
        for (uint32_t i = 0; i < EMB_MA; ++i) {
#pragma unroll
            for (uint32_t q = 0; q < 16; ++q) {
                w.acc[i][0][q] = __builtin_fmaf(w.acc[i][0][q], 0.5f, bias0);  // x 0.5 is exact: one rounding either way
                w.acc[i][1][q] = __builtin_fmaf(w.acc[i][1][q], 0.5f, bias1);
            }
        }
        if (a.relu) {
#pragma unroll
            for (uint32_t i = 0; i < EMB_MA; ++i) {
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) {
                    w.acc[i][0][q] = __builtin_amdgcn_fmed3f(w.acc[i][0][q], 0.0f, __builtin_inff());  // max(x, 0) in one op
                    w.acc[i][1][q] = __builtin_amdgcn_fmed3f(w.acc[i][1][q], 0.0f, __builtin_inff());
                }
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < EMB_MA; ++i) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                uint32_t d[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const f32x2 v = {w.acc[i][0][4u * j + k], w.acc[i][1][4u * j + k]};
                    d[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                }
                emb_quad_swap<0xB1>(d[0], d[1], (lane & 1u) != 0);  // lanes l ^ 1
                emb_quad_swap<0xB1>(d[2], d[3], (lane & 1u) != 0);
                emb_quad_swap<0x4E>(d[0], d[2], (lane & 2u) != 0);  // lanes l ^ 2
                emb_quad_swap<0x4E>(d[1], d[3], (lane & 2u) != 0);
                // now d[k] = the dword of quad lane k for row (lane & 3): columns 8 (r >> 2) + 2k, + 1 of this slab
                pend[4u * i + j] = make_uint4(d[0], d[1], d[2], d[3]);
            }
        }
        pend_rows = a.B - env0 < EMB_WAVE_ENVS ? (uint32_t)(a.B - env0) : EMB_WAVE_ENVS;
        pend_ptr = out_lane + env0 * ldw;
        env0 += pass_stride;
        if (env0 >= a.B) break;
#pragma unroll
        for (uint32_t i = 0; i < EMB_MA; ++i) pc[i] = pnx[i];
    }
#pragma unroll
    for (uint32_t qi = 0; qi < NQ; ++qi) store_quad(qi);
}


// Small batches (the reference collects 1 024 episodes at a time, rl/configs.py:134): embed_bits_kernel's workgroup passes 512 envs under
// a 64-column slab it first loads into LDS, so 1 024 envs are 16 workgroups x 256 MFMAs per wave -- 14 us on 16 CUs.  Here two waves
// own one tile of 32 envs x one slab, one per column of the slab's column pairs, and take their fragments from L2 straight into
// registers (the packed layout is fragment order: 1 KiB per wave load, EMS_AHEAD groups of 8 k-steps in flight).  What bounds these
// launches is the bytes a CU pulls, so every (tile, slab) pair is a workgroup of its own and the launch spreads over all CUs.
// Same packed weights, same expansion, same k order: bit-identical activations.
constexpr uint32_t EMS_AHEAD = 4;   // groups of 8 weight fragments in flight per wave
constexpr uint32_t EMS_WGS_PER_CU = 4;  // the launch takes this shape up to that many workgroups per CU

// one env tile per workgroup, two waves: one per column of the slab's column pairs
template <uint32_t G>
__global__ __launch_bounds__(128) void embed_small_kernel(EmbedArgs a) {
    constexpr uint32_t GP = (G + 1u) & ~1u;
    constexpr uint32_t slab_vec = GP * 8u * 2u * 64u;  // uint4 per slab (the padding group of an odd G has zero weights: skipped)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t slab = blockIdx.x % a.n_slabs;
    const uint32_t nb = wave & 1u;  // this wave's column of every pair: 2 c + nb
    const uint64_t tile = blockIdx.x / a.n_slabs;
    if (tile * 32u >= a.B) return;  // wave-uniform; no barrier below
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t sh[2] = {4u * h, 8u + 4u * h};
    uint64_t env = tile * 32u + r;
    env = env < a.B ? env : a.B - 1;  // tail: duplicate the last env, its rows are not stored
    const uint4 *ps = a.state + (env >> 6) * (uint64_t)(a.tile_groups * 64u) + (env & 63u) + (a.region ? (a.region[env] & 1u) * (G * 64u) : 0u);
    const uint4 *pw = a.wp + (uint64_t)slab * slab_vec + nb * 64u + lane;  // fragment (k-step s, nb) = pw[2 s * 64]
    uint4 bits[G], wb[EMS_AHEAD][8];
    auto fetch = [&](uint32_t g, uint4 (&b)[8]) {
#pragma unroll
        for (uint32_t ss = 0; ss < 8; ++ss) b[ss] = pw[((8u * g + ss) * 2u) * 64u];
    };
#pragma unroll
    for (uint32_t g = 0; g + 1u < EMS_AHEAD && g < G; ++g) fetch(g, wb[g]);
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) bits[g] = ps[g * 64u];
    const uint32_t n0 = slab * EMB_SLAB + 2u * r + nb;
    const float bias0 = a.bias ? a.bias[n0] : 0.0f;
    if (a.obs && slab == 0 && nb == 0 && tile * 32u + r < a.B) {
        // The rollout's packed observation from the bits this wave holds anyway (qm_pack_kernel's rows: CliffordEnv X-type row j = slot
        // 2j, Z-type row N + j = slot 2j + 1; LinearFunctionEnv row j = slot j).  Both lane halves hold the env: half h writes the
        // X (h = 0) / Z (h = 1) rows, or the even / odd groups of a LinearFunctionEnv.
        uint32_t *o = a.obs + (tile * 32u + r) * (uint64_t)a.D;
        if (a.has_z) {
            uint32_t *oh = o + (h ? a.N : 0u);
            if ((a.N & 3u) == 0) {  // whole 16-byte stores (the buffer is 16-byte aligned: the launcher checks)
#pragma unroll
                for (uint32_t g = 0; g + 1u < G; g += 2u)
                    if (2u * g < a.N)
                        *reinterpret_cast<uint4 *>(oh + 2u * g) = h ? make_uint4(bits[g].y, bits[g].w, bits[g + 1u].y, bits[g + 1u].w)
                                                                    : make_uint4(bits[g].x, bits[g].z, bits[g + 1u].x, bits[g + 1u].z);
            } else {
#pragma unroll
                for (uint32_t g = 0; g < G; ++g) {
                    if (2u * g < a.N) oh[2u * g] = h ? bits[g].y : bits[g].x;
                    if (2u * g + 1u < a.N) oh[2u * g + 1u] = h ? bits[g].w : bits[g].z;
                }
            }
        } else {
#pragma unroll
            for (uint32_t g = 0; g < G; ++g) {
                if ((g & 1u) != h) continue;
                const uint32_t w[4] = {bits[g].x, bits[g].y, bits[g].z, bits[g].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (4u * g + k < a.N) o[4u * g + k] = w[k];
            }
        }
    }
    f32x16 acc0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) acc0[q] = 0.0f;
#pragma unroll
    for (uint32_t g = 0; g < G; ++g) {
        if (g + EMS_AHEAD - 1u < G) fetch(g + EMS_AHEAD - 1u, wb[(g + EMS_AHEAD - 1u) % EMS_AHEAD]);
#pragma unroll
        for (uint32_t ss = 0; ss < 8; ++ss) {
            const uint32_t comp = ss >> 1;
            const uint32_t word = comp == 0 ? bits[g].x : comp == 1 ? bits[g].y : comp == 2 ? bits[g].z : bits[g].w;
            const bf16x8 af = emb_expand(word, sh[ss & 1u]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, wb[g % EMS_AHEAD][ss]), acc0, 0, 0, 0);
        }
    }
    // x 0.5 + bias (exact scaling: one rounding), ReLU, bf16; C layout: column 2 (lane & 31) + nb of this slab, env row = (q & 3) + 8 (q >> 2) + 4 h
    __hip_bfloat16 *out = reinterpret_cast<__hip_bfloat16 *>(a.out) + n0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
        float v0 = __builtin_fmaf(acc0[q], 0.5f, bias0);
        if (a.relu) v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, __builtin_inff());
        const uint64_t e = tile * 32u + (q & 3u) + 8u * (q >> 2) + 4u * h;
        if (e < a.B) out[e * a.ld_out] = __float2bfloat16(v0);
    }
}


// =================================================================================================
// Policy head + sampling in one kernel: logits = h W^T + b never leave the registers.
//
// The collector's last layer is Linear(256 -> num_actions) plus the value head (rl/configs.py:531-607), then a
// categorical draw per env.  As separate steps (hipBLASLt GEMM on a 171-column output, then qg_sample_actions
// re-reading the logits) they cost 21 + 26 us at 65 536 envs for 6 GFLOP.  Here the product is computed TRANSPOSED,
// logits^T = W h^T: the MFMA's A operand is a 32-action tile of W (from LDS, fragment order), the B operand 32
// envs of h (16 B per lane straight from global memory), so in the C layout a lane holds 16 actions per tile OF ITS
// OWN env (column = lane & 31 = env, row = action).  Max, sum of exponentials, entropy and the exponential race
// (the very draws of qg_sample_actions: same counter RNG, same hash, same tie rule) then run down the lane's
// registers without cross-lane traffic; the two lane halves, which hold complementary actions, meet once.
// The bias rides in an extra k-step (A = {bias_hi, bias_lo, 0...}, B = {1, 1, 0...}), padding rows carry bias
// -1e30 (exp -> 0, never win), the value head sits in the last padded row.
// =================================================================================================
constexpr uint32_t HEAD_WAVES = 8;
constexpr float HEAD_PAD_BIAS = -1.0e30f;

// Packed head: [k-step s <= K/16][tile t][lane][8] bf16; lane (r, h) element e of k-step s < K/16 = W[row(32t + r)][16s + 8h + e];
// k-step K/16: element 0 / 1 of the h = 0 lanes = bias hi / lo.  row(p) = p for p < A, value_row for p = 32 tiles - 1, else padding.
// xorder: the k order of a fragment built from an MFMA accumulator tile (mid_head_sample_kernel): element e of lane half h in
// k-step s is feature 32 (s >> 1) + 16 (s & 1) + 8 (e >> 2) + 4 h + (e & 3).  ks_total >= K/16 + 1 k-steps are written (zeros past the bias step).
template <typename WT>
__global__ __launch_bounds__(256) void pack_head_kernel(const WT *w, const WT *bias, uint64_t ld, uint32_t K, uint32_t A, int32_t value_row,
                                                        uint32_t tiles, uint32_t xorder, uint32_t ks_total, __hip_bfloat16 *out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ks = K / 16u;
    const uint64_t total = (uint64_t)ks_total * tiles * 64u * 8u;
    if (idx >= total) return;
    const uint32_t e = (uint32_t)idx & 7u, lane = (uint32_t)(idx >> 3) & 63u;
    const uint32_t st = (uint32_t)(idx >> 9), t = st % tiles, s = st / tiles;
    const uint32_t r = lane & 31u, h = lane >> 5, p = 32u * t + r;
    const int32_t src = p < A ? (int32_t)p : (p == 32u * tiles - 1u ? value_row : -1);
    float v = 0.0f;
    if (s < ks) {
        const uint32_t k = xorder ? 32u * (s >> 1) + 16u * (s & 1u) + 8u * (e >> 2) + 4u * h + (e & 3u) : 16u * s + 8u * h + e;
        if (src >= 0) v = (float)w[(uint64_t)src * ld + k];
    } else if (s == ks && h == 0 && e < 2) {
        const float b = src >= 0 ? (bias ? (float)bias[src] : 0.0f) : HEAD_PAD_BIAS;
        const float hi = __bfloat162float(__float2bfloat16(b));
        v = e == 0 ? hi : b - hi;
    }
    out[idx] = __float2bfloat16(v);
}

struct HeadArgs {
    const uint4 *h;        // [B][ld_h] bf16 activations
    const uint4 *wp;       // packed head
    void *actions;
    float *logp, *entropy, *values;
    const uint64_t *clock;
    uint64_t ld_h;         // elements per row of h
    uint64_t B, seed, counter;
    uint64_t env_base;     // global index of env 0 in the draw (the stepped handle's qg_vec_set_env_base; 0 without a handle)
    uint32_t K, A;
    int32_t act64;
};

__device__ __forceinline__ float head_xhalf(float x) { return __shfl_xor(x, 32, 64); }

// One logit of the draw as a function (mid_head_small_kernel; head_draw below spells the same operations out in its loop, whose
// register allocation the big kernels are tuned around): softmax terms relative to the env's maximum m and the exponential race of
// qg_sample_actions (kernels_collect.hip).
struct DrawLane {
    float best_q = __builtin_huge_valf(), best_d = 0.0f, ssum = 0.0f, wsum = 0.0f;
    uint32_t best_a = 0xFFFFFFFFu;
};
__device__ __forceinline__ void draw_logit(DrawLane &r, float logit, float m, uint32_t act, uint32_t bhi, uint32_t blo) {
    const float d = logit - m;
    const float ex = __builtin_amdgcn_exp2f(d * 1.44269504088896340736f);  // raw v_exp_f32: d <= 0, a flushed tiny result is 0 either way
    r.ssum += ex;
    r.wsum = __builtin_fmaf(ex, d, r.wsum);
    uint32_t x = bhi + act * 0x9E3779B9u;  // sample_uniform(base, act) (kernels_collect.hip)
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= blo;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    // u = ((x >> 9) + 0.5) 2^-23, built as (1 + (x >> 9) 2^-23) - (1 - 2^-24): both steps exact, bit-identical to sample_uniform
    const float u = __uint_as_float((x >> 9) | 0x3F800000u) - 0.99999994039535522461f;
    // the race compares -log(u) / ex; log2 instead of ln scales every key by the same positive constant
    const float qv = -__builtin_amdgcn_logf(u) * __builtin_amdgcn_rcpf(ex);  // padding rows: ex = 0, q = inf, never wins
    const bool take = qv < r.best_q;  // ascending action order within the lane: ties keep the lower index
    r.best_q = take ? qv : r.best_q;
    r.best_a = take ? act : r.best_a;
    r.best_d = take ? d : r.best_d;
}

// The draw from register-resident logits: acc[t][q] = logit of action 32 t + (q & 3) + 8 (q >> 2) + 4 h of env `env` (C layout of the
// transposed product), the last padded row is the value head.  Same race as qg_sample_actions (kernels_collect.hip).
template <uint32_t TILES>
__device__ __forceinline__ int64_t head_draw(f32x16 (&acc)[TILES], const HeadArgs &a, uint64_t env, bool live, uint32_t h) {
    const float INF = __builtin_huge_valf();
    // acc[t][q] = logit of action 32 t + (q & 3) + 8 (q >> 2) + 4 h for env `env`; the last padded row is the value head
    const float value = acc[TILES - 1][15];  // meaningful on the h = 1 lanes
    if (h == 1) acc[TILES - 1][15] = HEAD_PAD_BIAS;
    float m = -INF;
#pragma unroll
    for (uint32_t t = 0; t < TILES; ++t)
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) m = fmaxf(m, acc[t][q]);
    m = fmaxf(m, head_xhalf(m));
    const uint64_t base = rng_draw(a.seed, a.env_base + env, a.counter + clock_of(a.clock));
    const uint32_t blo = (uint32_t)base, xb = (uint32_t)(base >> 32) + 4u * h * 0x9E3779B9u;  // hash input of this lane half's action 0
    float best_q = INF, best_d = 0.0f, ssum = 0.0f, wsum = 0.0f;
    uint32_t best_a = 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t t = 0; t < TILES; ++t) {
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) {
            const uint32_t act = 32u * t + (q & 3u) + 8u * (q >> 2) + 4u * h;
            const float d = acc[t][q] - m;
            const float ex = __builtin_amdgcn_exp2f(d * 1.44269504088896340736f);  // raw v_exp_f32: d <= 0, a flushed tiny result is 0 either way
            ssum += ex;
            wsum = __builtin_fmaf(ex, d, wsum);
            uint32_t x = xb + (32u * t + (q & 3u) + 8u * (q >> 2)) * 0x9E3779B9u;  // sample_uniform(base, act) (kernels_collect.hip)
            x ^= x >> 16;
            x *= 0x7FEB352Du;
            x ^= blo;
            x ^= x >> 15;
            x *= 0x846CA68Bu;
            x ^= x >> 16;
            // u = ((x >> 9) + 0.5) 2^-23, built as (1 + (x >> 9) 2^-23) - (1 - 2^-24): both steps exact, bit-identical to sample_uniform
            const float u = __uint_as_float((x >> 9) | 0x3F800000u) - 0.99999994039535522461f;
            // the race compares -log(u) / ex; log2 instead of ln scales every key by the same positive constant
            const float qv = -__builtin_amdgcn_logf(u) * __builtin_amdgcn_rcpf(ex);  // padding rows: ex = 0, q = inf, never wins
            const bool take = qv < best_q;  // ascending action order within the lane: ties keep the lower index
            best_q = take ? qv : best_q;
            best_a = take ? act : best_a;
            best_d = take ? d : best_d;
        }
    }
    {   // the other lane half holds the other actions of this env
        const float oq = head_xhalf(best_q), od = head_xhalf(best_d);
        const uint32_t oa = __shfl_xor(best_a, 32, 64);
        ssum += head_xhalf(ssum);
        wsum += head_xhalf(wsum);
        const bool take = oa != 0xFFFFFFFFu && (best_a == 0xFFFFFFFFu || oq < best_q || (oq == best_q && oa < best_a));
        best_q = take ? oq : best_q;
        best_a = take ? oa : best_a;
        best_d = take ? od : best_d;
    }
    const float v_other = head_xhalf(value);
    const int64_t act = best_a == 0xFFFFFFFFu ? 0 : (int64_t)best_a;  // the env's action on both lane halves
    if (live && h == 0) {
        if (a.act64) reinterpret_cast<int64_t *>(a.actions)[env] = act;
        else reinterpret_cast<int32_t *>(a.actions)[env] = (int32_t)act;
        const float log_s = logf(ssum);
        if (a.logp) a.logp[env] = best_d - log_s;
        if (a.entropy) a.entropy[env] = log_s - wsum / ssum;
        if (a.values) a.values[env] = v_other;
    }
    return act;
}

template <uint32_t TILES>
__global__ __launch_bounds__(64 * HEAD_WAVES, 1) void head_sample_kernel(HeadArgs a) {
    extern __shared__ uint4 head_lds[];  // packed head: [(K/16 + 1)][TILES][64]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ks = a.K / 16u;
    const uint32_t total_vec = (ks + 1u) * TILES * 64u;
    for (uint32_t c = wave * 64u; c < total_vec; c += 64u * HEAD_WAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.wp + c + lane),
                                         (__attribute__((address_space(3))) void *)(head_lds + c), 16, 0, 0);
    const uint32_t c = lane & 31u, h = lane >> 5;
    const uint64_t n_tiles = (a.B + 31u) / 32u;
    const uint64_t row_vec = a.ld_h / 8u;  // uint4 per row of h
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint64_t tile = (uint64_t)blockIdx.x * HEAD_WAVES + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * HEAD_WAVES) {
        const uint64_t env_raw = tile * 32u + c;
        const bool live = env_raw < a.B;
        const uint64_t env = live ? env_raw : a.B - 1;
        const uint4 *hrow = a.h + env * row_vec + h;  // k-step s: 16 B at element 16 s + 8 h
        f32x16 acc[TILES];
#pragma unroll
        for (uint32_t t = 0; t < TILES; ++t)
#pragma unroll
            for (uint32_t q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        // four k-steps per trip, the next trip's activations in flight meanwhile
        uint4 bq[4], bn[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) bq[j] = hrow[2u * j];
        for (uint32_t s0 = 0; s0 < ks; s0 += 4u) {
            const uint32_t sn = s0 + 4u < ks ? s0 + 4u : s0;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) bn[j] = hrow[2u * (sn + j)];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const bf16x8 bf = __builtin_bit_cast(bf16x8, bq[j]);
                const uint4 *al = head_lds + (uint64_t)(s0 + j) * (TILES * 64u) + lane;
#pragma unroll
                for (uint32_t t = 0; t < TILES; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[t * 64u]), bf, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) bq[j] = bn[j];
        }
        {   // the bias k-step: B = {1, 1, 0, ...} on the k-half-0 lanes
            const uint4 ones = make_uint4(h == 0 ? 0x3F803F80u : 0u, 0u, 0u, 0u);
            const bf16x8 bf = __builtin_bit_cast(bf16x8, ones);
            const uint4 *al = head_lds + (uint64_t)ks * (TILES * 64u) + lane;
#pragma unroll
            for (uint32_t t = 0; t < TILES; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[t * 64u]), bf, acc[t], 0, 0, 0);
        }
        head_draw<TILES>(acc, a, env, live, h);
    }
}


// =================================================================================================
// Middle layer + head + draw in one kernel: h2 = relu(h1 W2^T + b2) never leaves the registers either.
//
// Both products are computed transposed.  X = W2 h1^T: A operand = a 32-feature tile of W2 (LDS), B operand = 32 envs of
// h1 (16 B per lane from global memory); the accumulator tile X[t] then has its 32 features in the registers and the
// env on the lane -- exactly what the head's B operand needs, because the head sums over the features (the guide's
// "accumulator tile as the next MFMA's operand": registers 8s..8s+7 packed to bf16 are the fragment of k-step 2t + s,
// in a permuted k order that the head's packed weights follow, `xorder` above).  Biases ride in extra k-steps.
//
// Nothing is LDS-resident: W2 (256 KiB) and then the head (<= 102 KiB) stream through two 16 KiB buffers in chunks of
// 16 fragments, shared by the workgroup's 4 waves (one barrier per chunk).  With 32 KiB of LDS and one wave per SIMD a
// CU holds two workgroups (39 -> 36 us at 65 536 envs against one 8-wave workgroup with the head resident; starting the
// second workgroup of a CU one MFMA phase late, so that its MFMAs would run under the first one's draw, measured no gain).
// =================================================================================================
constexpr uint32_t MID_FT = 8;        // 32-feature tiles of the middle layer (256 features)
constexpr uint32_t MID_CHUNK = 2;     // k-steps of W2 per streamed chunk: MID_CHUNK * MID_FT = 16 fragments of 1 KiB
constexpr uint32_t MH_WAVES = 4;
constexpr uint32_t MH_CHUNK_FRAGS = MID_CHUNK * MID_FT;
constexpr uint32_t MH_CHUNK_VEC = MH_CHUNK_FRAGS * 64u;  // uint4 per chunk

struct MidHeadArgs {
    HeadArgs head;         // head.h = h1, head.ld_h its stride, head.K = features of the middle layer (256), head.wp = packed head (xorder)
    const uint4 *w2p;      // packed middle layer: [K1/16 + 2 k-steps][MID_FT][64]
    uint32_t K1;           // in_features of the middle layer
    // qg_vec_mid_head_sample_step: Env::step with the drawn action in the same launch (TILE layout, no add_inverts), and the indices
    // of the envs whose episode ended appended to the handle's list for the next qg_vec_reset_done.  step.state == nullptr: draw only.
    StepArgs step;
    uint32_t step_groups;  // 16-byte groups per env of the TILE layout
    uint32_t step_has_z;
    uint32_t *done_list, *done_count;
    // auto-reset (mid_head_small_kernel): the envs whose episode ended in this step start the next one before the launch ends --
    // qg_vec_reset_done(reset.seed) without its own launch.  reset.state == nullptr: no reset (the done list is appended instead).
    InitArgs reset;
};

// TPW = env tiles per wave.  TPW = 1: two workgroups per CU (the second one's MFMA phase can run under the first one's draw).  TPW = 2: one
// workgroup per CU whose waves carry two tiles each -- every weight fragment read from LDS feeds two MFMAs, and a tile pays half the chunk
// barriers, staging instructions and LDS reads; the accumulators of both tiles (256 + 136 + 32 TILES registers at the widest point) need
// the whole register file of a SIMD, i.e. one wave per SIMD.
template <uint32_t TILES, uint32_t TPW>
__global__ __launch_bounds__(64 * MH_WAVES, TPW == 1 ? 2 : 1) void mid_head_sample_kernel(MidHeadArgs ma) {
    __shared__ uint4 cbuf[2 * MH_CHUNK_VEC];
    const HeadArgs &a = ma.head;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t HEAD_FRAGS = (2u * MID_FT + 1u) * TILES;                       // k-steps 0..16 (16 = bias) x action tiles
    constexpr uint32_t HEAD_CHUNKS = (HEAD_FRAGS + MH_CHUNK_FRAGS - 1u) / MH_CHUNK_FRAGS;
    const uint32_t n2 = ma.K1 / (16u * MID_CHUNK) + 1u;                               // W2 chunks; the last is {bias k-step, zero k-step}
    auto stage = [&](uint32_t sidx) {  // stream chunk sidx -> buffer sidx & 1 (4 wave-instructions per wave)
        const uint4 *src = sidx < n2 ? ma.w2p + (uint64_t)sidx * MH_CHUNK_VEC : a.wp + (uint64_t)(sidx - n2) * MH_CHUNK_VEC;
        uint4 *dst = cbuf + (sidx & 1u) * MH_CHUNK_VEC;
#pragma unroll
        for (uint32_t c = 0; c < MH_CHUNK_VEC; c += 64u * MH_WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c + wave * 64u + lane),
                                             (__attribute__((address_space(3))) void *)(dst + c + wave * 64u), 16, 0, 0);
    };
    auto landed = [&]() {  // every wave's pieces of the chunk in flight have landed and every wave is done with the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const uint32_t c = lane & 31u, h = lane >> 5;
    const uint64_t n_tiles = (a.B + 31u) / 32u;
    const uint64_t row_vec = a.ld_h / 8u;
    const uint64_t tiles_per_trip = (uint64_t)gridDim.x * MH_WAVES * TPW;
    const uint4 ones = make_uint4(h == 0 ? 0x3F803F80u : 0u, 0u, 0u, 0u), zeros = make_uint4(0u, 0u, 0u, 0u);
    // every wave of the workgroup takes part in every trip (the chunk barriers): waves past the last tile compute on a clamped env
    for (uint64_t tile0 = (uint64_t)blockIdx.x * MH_WAVES * TPW; tile0 < n_tiles; tile0 += tiles_per_trip) {
        bool live[TPW];
        uint64_t env[TPW];
        const uint4 *hrow[TPW];
#pragma unroll
        for (uint32_t p = 0; p < TPW; ++p) {
            const uint64_t env_raw = (tile0 + wave * TPW + p) * 32u + c;
            live[p] = env_raw < a.B;
            env[p] = live[p] ? env_raw : a.B - 1;
            hrow[p] = a.h + env[p] * row_vec + h;
        }
        stage(0);
        uint4 bq[TPW][MID_CHUNK], bn[TPW][MID_CHUNK];
#pragma unroll
        for (uint32_t p = 0; p < TPW; ++p)
#pragma unroll
            for (uint32_t j = 0; j < MID_CHUNK; ++j) bq[p][j] = hrow[p][2u * j];
        f32x16 x[TPW][MID_FT];
#pragma unroll
        for (uint32_t p = 0; p < TPW; ++p)
#pragma unroll
            for (uint32_t t = 0; t < MID_FT; ++t)
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) x[p][t][q] = 0.0f;
        landed();
        for (uint32_t ch = 0; ch < n2; ++ch) {
            stage(ch + 1u);  // W2's next chunk, or the head's first; its buffer was last read in chunk ch - 1, which every wave has left
#pragma unroll
            for (uint32_t p = 0; p < TPW; ++p) {
                if (ch + 2u < n2) {
#pragma unroll
                    for (uint32_t j = 0; j < MID_CHUNK; ++j) bn[p][j] = hrow[p][2u * ((ch + 1u) * MID_CHUNK + j)];
                } else {
                    bn[p][0] = ones;
                    bn[p][1] = zeros;
                }
            }
            const uint4 *al = cbuf + (ch & 1u) * MH_CHUNK_VEC + lane;
#pragma unroll
            for (uint32_t j = 0; j < MID_CHUNK; ++j) {
#pragma unroll
                for (uint32_t t = 0; t < MID_FT; ++t) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, al[(j * MID_FT + t) * 64u]);  // one LDS read, TPW MFMAs
#pragma unroll
                    for (uint32_t p = 0; p < TPW; ++p)
                        x[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bq[p][j]), x[p][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (uint32_t p = 0; p < TPW; ++p)
#pragma unroll
                for (uint32_t j = 0; j < MID_CHUNK; ++j) bq[p][j] = bn[p][j];
            landed();
        }
        // ReLU, bf16: the tiles become the head's B fragments, registers 8s..8s+7 of tile t = k-step 2t + s; k-step 16 = bias
        bf16x8 hb[TPW][2u * MID_FT + 1u];
#pragma unroll
        for (uint32_t p = 0; p < TPW; ++p) {
#pragma unroll
            for (uint32_t t = 0; t < MID_FT; ++t) {
#pragma unroll
                for (uint32_t s = 0; s < 2; ++s) {
                    u32x4 f;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const f32x2 v = {__builtin_amdgcn_fmed3f(x[p][t][8u * s + 2u * j], 0.0f, __builtin_inff()),
                                         __builtin_amdgcn_fmed3f(x[p][t][8u * s + 2u * j + 1u], 0.0f, __builtin_inff())};
                        f[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                    }
                    hb[p][2u * t + s] = __builtin_bit_cast(bf16x8, f);
                }
            }
            hb[p][2u * MID_FT] = __builtin_bit_cast(bf16x8, ones);
        }
        f32x16 acc[TPW][TILES];
#pragma unroll
        for (uint32_t p = 0; p < TPW; ++p)
#pragma unroll
            for (uint32_t t = 0; t < TILES; ++t)
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) acc[p][t][q] = 0.0f;
        // the head: fragment f = k-step (f / TILES) x action tile (f % TILES), 16 fragments per streamed chunk
#pragma unroll
        for (uint32_t hc = 0; hc < HEAD_CHUNKS; ++hc) {
            if (hc + 1u < HEAD_CHUNKS) stage(n2 + hc + 1u);
            const uint4 *al = cbuf + ((n2 + hc) & 1u) * MH_CHUNK_VEC + lane;
#pragma unroll
            for (uint32_t i = 0; i < MH_CHUNK_FRAGS; ++i) {
                const uint32_t f = hc * MH_CHUNK_FRAGS + i;
                if (f < HEAD_FRAGS) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, al[i * 64u]);
#pragma unroll
                    for (uint32_t p = 0; p < TPW; ++p)
                        acc[p][f % TILES] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, hb[p][f / TILES], acc[p][f % TILES], 0, 0, 0);
                }
            }
            landed();  // after the last chunk: every wave is done with both buffers before the next trip stages chunk 0
        }
#pragma unroll
        for (uint32_t p = 0; p < TPW; ++p) {
            const int64_t act = head_draw<TILES>(acc[p], a, env[p], live[p], h);
            if (ma.step.state) {  // wave-uniform
                bool fin = false;
                if (live[p] && h == 0)
                    fin = ma.step_has_z ? qm_step1_body<true, true>(ma.step, ma.step_groups, env[p], act) : qm_step1_body<false, true>(ma.step, ma.step_groups, env[p], act);
                // compact_done (kernels_collect.hip) for this wave's 32 envs: one atomic per wave with a finished env
                const uint64_t m = __ballot(fin);
                if (m) {
                    const uint32_t first = (uint32_t)__ffsll((long long)m) - 1u;
                    uint32_t base = 0;
                    if (lane == first) base = atomicAdd(ma.done_count, (uint32_t)__popcll(m));
                    base = __shfl(base, first);
                    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (fin && slot < ma.step.B) ma.done_list[slot] = (uint32_t)env[p];
                }
            }
        }
    }
}


// Small batches (the reference collects 1 024 episodes at a time, rl/configs.py:134): mid_head_sample_kernel gives a workgroup 128 envs
// and lets it stream all 358 KB of weights, so 1 024 envs occupy 8 CUs for 23 us.  Here a workgroup's four waves share ONE tile of 32
// envs: wave w multiplies feature tiles 2w, 2w + 1 of the middle layer and action tiles w, w + 4 of the head, so nobody shares a
// weight fragment and each goes from L2 straight into registers (the packed layouts are already in fragment order: 1 KiB per wave
// load), three groups of 8 k-steps ahead; the tile's activations are loaded once and shared through LDS.  h2 crosses the waves through 16 KiB of LDS, the draw runs on a quarter of the logits per wave with the
// env's maximum and the partial sums / race winners exchanged through LDS, wave 0 finishes (and steps the env).  Same packed weights,
// same k order, same keys: actions are those of mid_head_sample_kernel; logp / entropy sum their terms in another order.
constexpr uint32_t MHS_WAVES = 4;
constexpr uint32_t MHS_GROUP = 8;   // k-steps per register group of the middle layer
constexpr uint32_t MHS_KMAX = 64;   // k-steps of activations the workgroup keeps in LDS (in_features <= 1024)

// Workgroup barrier for data exchanged through LDS only: __syncthreads() fences every address space, which also drains the vector-memory
// counter and with it the weight fragments requested ahead.
__device__ __forceinline__ void mhs_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <uint32_t TILES>
__global__ __launch_bounds__(64 * MHS_WAVES, 1) void mid_head_small_kernel(MidHeadArgs ma) {
    static_assert(MID_FT == 2 * MHS_WAVES && TILES <= 2 * MHS_WAVES, "two feature tiles and <= two action tiles per wave");
    __shared__ uint4 bbuf[MHS_KMAX * 64u];          // the tile's activations as B fragments, k-step major: loaded once, read by all four waves
    __shared__ uint4 hbuf[2u * MID_FT * 64u];       // h2 as the head's B fragments, k-step major
    __shared__ float xmax[MHS_WAVES][32];
    __shared__ float part[MHS_WAVES][6][32];         // ssum, wsum, best_q, best_d, best_a, value
    const HeadArgs &a = ma.head;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t c = lane & 31u, h = lane >> 5;
    const uint64_t env_raw = (uint64_t)blockIdx.x * 32u + c;
    const bool live = env_raw < a.B;
    const uint64_t env = live ? env_raw : a.B - 1;
    const uint4 *hrow = a.h + env * (a.ld_h / 8u) + h;
    const uint4 ones = make_uint4(h == 0 ? 0x3F803F80u : 0u, 0u, 0u, 0u);
    const uint32_t ks1 = ma.K1 / 16u;  // real k-steps of the middle layer (a multiple of MHS_GROUP, <= MHS_KMAX: the launcher checks), then the bias k-step
    const uint32_t ng = ks1 / MHS_GROUP;
    // ---- middle layer: x[i] = feature tile 2 wave + i ----
    const uint4 *wa = ma.w2p + (2u * wave) * 64u + lane;  // fragment (k-step s, tile 2 wave + i) = wa[(s * MID_FT + i) * 64]
    f32x16 x[2];
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i)
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) x[i][q] = 0.0f;
    uint4 fa[3][MHS_GROUP][2];  // three groups of weight fragments in flight
    auto fetch = [&](uint32_t g, uint4 (&A)[MHS_GROUP][2]) {
#pragma unroll
        for (uint32_t j = 0; j < MHS_GROUP; ++j) {
            A[j][0] = wa[((g * MHS_GROUP + j) * MID_FT) * 64u];
            A[j][1] = wa[((g * MHS_GROUP + j) * MID_FT + 1u) * 64u];
        }
    };
    auto multiply = [&](uint32_t g, const uint4 (&A)[MHS_GROUP][2]) {
#pragma unroll
        for (uint32_t j = 0; j < MHS_GROUP; ++j) {
            const bf16x8 bf = __builtin_bit_cast(bf16x8, bbuf[(g * MHS_GROUP + j) * 64u + lane]);
            x[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[j][0]), bf, x[0], 0, 0, 0);
            x[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[j][1]), bf, x[1], 0, 0, 0);
        }
    };
    fetch(0, fa[0]);
    if (1u < ng) fetch(1, fa[1]);
    if (2u < ng) fetch(2, fa[2]);
    // the activations: wave w brings groups w, w + 4 (<= 2 x 8 k-steps, 16 B per lane each) and parks them in LDS for all four waves.
    // Requested after the weights: vector memory returns in order, so once they are here so are the three weight groups -- everything
    // the workgroup can have in flight is in flight from the first instruction on (the limit is what a CU's memory path delivers).
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i) {
        const uint32_t g = wave + MHS_WAVES * i;
        if (g < ng) {
#pragma unroll
            for (uint32_t j = 0; j < MHS_GROUP; ++j) bbuf[(g * MHS_GROUP + j) * 64u + lane] = hrow[2u * (g * MHS_GROUP + j)];
        }
    }
    mhs_lds_barrier();
    const uint4 bias0 = wa[(ks1 * MID_FT) * 64u], bias1 = wa[(ks1 * MID_FT + 1u) * 64u];
    // ---- the head's weights for this wave's action tiles (wave, wave + 4), all 17 k-steps: requested behind the
    // activations, they arrive while the middle layer multiplies ----
    const bool two = wave + MHS_WAVES < TILES, any = wave < TILES;  // wave-uniform
    const uint4 *wh = a.wp + wave * 64u + lane;  // fragment (k-step s, tile wave + 4 i) = wh[(s * TILES + 4 i) * 64]
    uint4 ha[2u * MID_FT + 1u][2];
#pragma unroll
    for (uint32_t s = 0; s <= 2u * MID_FT; ++s) {
        ha[s][0] = any ? wh[(s * TILES) * 64u] : make_uint4(0u, 0u, 0u, 0u);
        ha[s][1] = two ? wh[(s * TILES + MHS_WAVES) * 64u] : make_uint4(0u, 0u, 0u, 0u);
    }
    for (uint32_t g0 = 0; g0 < ng; g0 += 3u) {  // three groups per trip: the register buffers are indexed statically
        multiply(g0, fa[0]);
        if (g0 + 3u < ng) fetch(g0 + 3u, fa[0]);
        if (g0 + 1u < ng) {
            multiply(g0 + 1u, fa[1]);
            if (g0 + 4u < ng) fetch(g0 + 4u, fa[1]);
        }
        if (g0 + 2u < ng) {
            multiply(g0 + 2u, fa[2]);
            if (g0 + 5u < ng) fetch(g0 + 5u, fa[2]);
        }
    }
    {   // bias k-step of the middle layer: B = {1, 1, 0, ...} on the k-half-0 lanes
        const bf16x8 bf = __builtin_bit_cast(bf16x8, ones);
        x[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bias0), bf, x[0], 0, 0, 0);
        x[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bias1), bf, x[1], 0, 0, 0);
    }
    // ReLU, bf16: registers 8s..8s+7 of feature tile t = the head's k-step 2t + s (mid_head_sample_kernel)
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i) {
#pragma unroll
        for (uint32_t sft = 0; sft < 2; ++sft) {
            u32x4 f;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const f32x2 v = {__builtin_amdgcn_fmed3f(x[i][8u * sft + 2u * j], 0.0f, __builtin_inff()),
                                 __builtin_amdgcn_fmed3f(x[i][8u * sft + 2u * j + 1u], 0.0f, __builtin_inff())};
                f[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
            }
            hbuf[(2u * (2u * wave + i) + sft) * 64u + lane] = __builtin_bit_cast(uint4, f);
        }
    }
    mhs_lds_barrier();  // the head's fragments stay in flight
    // ---- head: acc[i] = action tile wave + 4 i ----
    f32x16 acc[2];
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i)
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) acc[i][q] = 0.0f;
    if (any) {
#pragma unroll
        for (uint32_t s = 0; s <= 2u * MID_FT; ++s) {
            const bf16x8 bf = __builtin_bit_cast(bf16x8, s < 2u * MID_FT ? hbuf[s * 64u + lane] : ones);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ha[s][0]), bf, acc[0], 0, 0, 0);
            if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ha[s][1]), bf, acc[1], 0, 0, 0);
        }
    }
    // ---- draw: this wave's logits are acc[i][q] = action 32 (wave + 4 i) + (q & 3) + 8 (q >> 2) + 4 h of env c ----
    const float INF = __builtin_huge_valf();
    constexpr uint32_t VW = (TILES - 1u) % MHS_WAVES, VI = (TILES - 1u) / MHS_WAVES;  // the value head: last row of the last tile (h = 1, q = 15)
    float value = 0.0f;
    if (wave == VW) {
        value = acc[VI][15];
        if (h == 1) acc[VI][15] = HEAD_PAD_BIAS;
    }
    float m = -INF;
    if (any) {
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) m = fmaxf(m, acc[0][q]);
        if (two) {
#pragma unroll
            for (uint32_t q = 0; q < 16; ++q) m = fmaxf(m, acc[1][q]);
        }
    }
    m = fmaxf(m, head_xhalf(m));
    if (h == 0) xmax[wave][c] = m;
    __syncthreads();
    m = fmaxf(fmaxf(xmax[0][c], xmax[1][c]), fmaxf(xmax[2][c], xmax[3][c]));
    const uint64_t base = rng_draw(a.seed, a.env_base + env, a.counter + clock_of(a.clock));
    const uint32_t blo = (uint32_t)base, bhi = (uint32_t)(base >> 32);
    DrawLane r;
    if (any) {
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) draw_logit(r, acc[0][q], m, 32u * wave + (q & 3u) + 8u * (q >> 2) + 4u * h, bhi, blo);
        if (two) {
#pragma unroll
            for (uint32_t q = 0; q < 16; ++q) draw_logit(r, acc[1][q], m, 32u * (wave + MHS_WAVES) + (q & 3u) + 8u * (q >> 2) + 4u * h, bhi, blo);
        }
    }
    {   // the other lane half holds the other actions of these tiles
        const float oq = head_xhalf(r.best_q), od = head_xhalf(r.best_d);
        const uint32_t oa = __shfl_xor(r.best_a, 32, 64);
        r.ssum += head_xhalf(r.ssum);
        r.wsum += head_xhalf(r.wsum);
        const bool take = oa != 0xFFFFFFFFu && (r.best_a == 0xFFFFFFFFu || oq < r.best_q || (oq == r.best_q && oa < r.best_a));
        r.best_q = take ? oq : r.best_q;
        r.best_a = take ? oa : r.best_a;
        r.best_d = take ? od : r.best_d;
    }
    const float v_other = head_xhalf(value);
    if (h == 0) {
        part[wave][0][c] = r.ssum;
        part[wave][1][c] = r.wsum;
        part[wave][2][c] = r.best_q;
        part[wave][3][c] = r.best_d;
        part[wave][4][c] = __uint_as_float(r.best_a);
        if (wave == VW) part[0][5][c] = v_other;
    }
    __syncthreads();
    if (wave != 0) return;
    float ssum = 0.0f, wsum = 0.0f, best_q = INF, best_d = 0.0f;
    uint32_t best_a = 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t w = 0; w < MHS_WAVES; ++w) {  // ascending action tiles within a wave, ties across waves by the lower index
        ssum += part[w][0][c];
        wsum += part[w][1][c];
        const float oq = part[w][2][c], od = part[w][3][c];
        const uint32_t oa = __float_as_uint(part[w][4][c]);
        const bool take = oa != 0xFFFFFFFFu && (best_a == 0xFFFFFFFFu || oq < best_q || (oq == best_q && oa < best_a));
        best_q = take ? oq : best_q;
        best_a = take ? oa : best_a;
        best_d = take ? od : best_d;
    }
    const int64_t act = best_a == 0xFFFFFFFFu ? 0 : (int64_t)best_a;
    if (live && h == 0) {
        if (a.act64) reinterpret_cast<int64_t *>(a.actions)[env] = act;
        else reinterpret_cast<int32_t *>(a.actions)[env] = (int32_t)act;
        const float log_s = logf(ssum);
        if (a.logp) a.logp[env] = best_d - log_s;
        if (a.entropy) a.entropy[env] = log_s - wsum / ssum;
        if (a.values) a.values[env] = part[0][5][c];
    }
    if (ma.step.state) {  // wave-uniform
        // `fin` on ONE lane per env (renv): with add_inverts the step is qm_inv2_body's, two adjacent lanes per env, so the 32 envs
        // move from lanes (c, c + 32) to lanes (2e, 2e + 1) first
        bool fin = false;
        uint64_t renv = env;
        if (ma.step.flags & F_INVERTS) {
            const uint32_t e = lane >> 1;
            const int64_t act2 = (int64_t)__shfl((int)act, (int)e);  // lanes 0 .. 31 hold the action of env c = lane
            const uint64_t env2 = (uint64_t)blockIdx.x * 32u + e;
            renv = env2;
            if (env2 < a.B) fin = qm_inv2_body<0, true>(ma.step, ma.step_groups, env2, lane & 1u, &act2) && !(lane & 1u);
        } else if (live && h == 0) {
            fin = ma.step_has_z ? qm_step1_body<true, true>(ma.step, ma.step_groups, env, act) : qm_step1_body<false, true>(ma.step, ma.step_groups, env, act);
        }
        if (ma.reset.state) {
            // qg_vec_reset_done for this wave's finished envs (clifford.rs:306-318; qm_init_kernel's mode 2 + qm_init_finish, kernels_qm.hip):
            // identity, `difficulty` random gates on LDS-resident rows ([slot][lane], scramble_flat), rows back to the tile, bookkeeping
            // of a fresh episode.
            if (fin) {
                const InitArgs &ia = ma.reset;
                uint32_t (*rows)[QG_WAVE] = reinterpret_cast<uint32_t (*)[QG_WAVE]>(bbuf);  // free since the middle layer; 4 G x 64 words
                const uint32_t R = 4u * ma.step_groups, N = ia.N;
                auto ident = [&](uint32_t k) -> uint32_t {
                    const uint32_t j = ma.step_has_z ? k >> 1 : k;
                    return j < N ? ((ma.step_has_z && (k & 1u)) ? (1u << N) << j : 1u << j) : 0u;
                };
                for (uint32_t k = 0; k < R; ++k) rows[k][lane] = ident(k);
                scramble_flat<uint32_t>(rows, lane, ia, renv);
                uint4 *tile = reinterpret_cast<uint4 *>(ia.state) + (renv >> 6) * (uint64_t)(ma.step_groups * 64u) + (renv & 63u);
                uint32_t bad = 0;
                for (uint32_t g = 0; g < ma.step_groups; ++g) {
                    const uint32_t w0 = rows[4u * g][lane], w1 = rows[4u * g + 1u][lane], w2 = rows[4u * g + 2u][lane], w3 = rows[4u * g + 3u][lane];
                    tile[g * 64u] = make_uint4(w0, w1, w2, w3);
                    if (ma.step_has_z) {  // bit j: qubit j's two rows differ from the identity's (qm_badmask)
                        bad |= (uint32_t)(w0 != ident(4u * g) || w1 != ident(4u * g + 1u)) << (2u * g);
                        bad |= (uint32_t)(w2 != ident(4u * g + 2u) || w3 != ident(4u * g + 3u)) << (2u * g + 1u);
                    } else {
                        bad |= (uint32_t)(w0 != ident(4u * g)) << (4u * g);
                        bad |= (uint32_t)(w1 != ident(4u * g + 1u)) << (4u * g + 1u);
                        bad |= (uint32_t)(w2 != ident(4u * g + 2u)) << (4u * g + 2u);
                        bad |= (uint32_t)(w3 != ident(4u * g + 3u)) << (4u * g + 3u);
                    }
                }
                const bool solved = bad == 0;
                if (ia.bad) ia.bad[renv] = bad;
                ia.depth[renv] = ia.depth_value;  // reset_internals (clifford.rs:272-283)
                ia.success[renv] = (uint8_t)solved;
                ia.reward[renv] = solved ? 1.0f : 0.0f;
                ia.done[renv] = (uint8_t)(ia.depth_value == 0 || solved);
                ia.inverted[renv] = ia.check_symplectic ? (uint8_t)QM_FLAG_SYMPLECTIC : (uint8_t)0;  // identity + gates is symplectic (qm_init_finish)
                ia.error[renv] = 0;
                ia.sol_len[renv * 2] = 0;
                ia.sol_len[renv * 2 + 1] = 0;
                if (ia.layers) {
                    const LayerRec lay = layer_rec(ia.layers, renv, ia.layers_len);
                    for (uint32_t i = 0; i + 2 < ia.layers_len; ++i) lay[i] = -1;
                    lay[ia.layers_len - 2] = 0;
                    lay[ia.layers_len - 1] = 0;
                }
            }
            return;
        }
        const uint64_t mk = __ballot(fin);
        if (mk) {
            const uint32_t first = (uint32_t)__ffsll((long long)mk) - 1u;
            uint32_t basei = 0;
            if (lane == first) basei = atomicAdd(ma.done_count, (uint32_t)__popcll(mk));
            basei = __shfl(basei, first);
            const uint32_t slot = basei + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
            if (fin && slot < ma.step.B) ma.done_list[slot] = (uint32_t)renv;
        }
    }
}


// =================================================================================================
// First layer from packed observation WORDS: any env kind, any rank's shard.
//
// embed_bits_kernel above reads the resident TILE layout and keeps a whole 64-column weight slab in LDS, which caps K
// at 1024.  The packed observation every layout exports (qg_vec_observe_packed: one 64-bit word per observation row --
// PauliEnv 2N rows of 2N + max_rotations columns, CliffordEnv N > 16, and what the multi-GPU all-gather moves) has
// K = 64 * rows bit positions, of which only `cols` per row carry weights.  Same bit trick for the A operand (a lane
// loads 16 B = two rows of its env and expands them in registers), but the weights stream: a workgroup of 4 waves owns
// 256 envs x 128 output columns, the packed weights of one 16-byte group (KPG k-steps x 4 fragments = KPG x 4 KiB)
// arrive per chunk through two LDS buffers (global_load_lds, one barrier per chunk), every A fragment feeds four
// B fragments (8 MFMAs per k-step per wave for 2 expansions), two workgroups per CU.  Only k-steps that can hold
// weights exist: a row's low word takes two, its high word two (cols > 48), one (32 < cols <= 48: the <= 16 valid
// bits are first moved to bits 0..7 / 16..23, which one rotation covers for both lane halves) or none (cols <= 32).
// =================================================================================================
constexpr uint32_t EW_WAVES = 4;
constexpr uint32_t EW_MA = 2;                       // 32-env row tiles per wave
constexpr uint32_t EW_NB = 4;                       // B fragments per k-step: 2 slabs of 64 columns x 2
constexpr uint32_t EW_COLS = 32u * EW_NB;           // output columns per workgroup
constexpr uint32_t EW_BLOCK_ENVS = EW_WAVES * 32u * EW_MA;

__host__ __device__ inline uint32_t ew_kpg(uint32_t cols) { return cols <= 32u ? 4u : cols <= 48u ? 6u : 8u; }  // k-steps per 16-byte group (two rows)

// k-step s of a group: which of the group's four 32-bit words {row0 lo, row0 hi, row1 lo, row1 hi} and which rotation:
// 0 / 1 = the two phases of embed_bits_kernel (bits {11..18, 27..31, 0..2} / {3..10, 19..26}), 2 = the compressed high word
__host__ __device__ inline void ew_kstep(uint32_t kpg, uint32_t s, uint32_t &word, uint32_t &mode) {
    if (kpg == 8u) { word = s >> 1; mode = s & 1u; }
    else if (kpg == 4u) { word = (s >> 1) * 2u; mode = s & 1u; }
    else { const uint32_t r = s / 3u, k = s % 3u; word = 2u * r + (k == 2u ? 1u : 0u); mode = k == 2u ? 2u : k; }
}

// Packed weights: [column tile of 128][group g < rows / 2][k-step s < KPG][fragment f < 4][lane][e < 8] bf16.  Fragment f of
// column tile ct holds, for lane (c = lane & 31, h = lane >> 5), output column ct * 128 + 64 (f >> 1) + 2 c + (f & 1); element
// e = 2 j + half multiplies the observation bit the kernel's A element (j, half) of that k-step is built from.
template <typename WT>
__global__ __launch_bounds__(256) void pack_embed_words_kernel(const WT *w, uint64_t ld, uint32_t hidden, uint32_t rows, uint32_t cols, __hip_bfloat16 *out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t kpg = ew_kpg(cols), G = rows / 2u;
    const uint64_t total = (uint64_t)(hidden / EW_COLS) * G * kpg * EW_NB * 64u * 8u;
    if (idx >= total) return;
    const uint32_t e = (uint32_t)idx & 7u, lane = (uint32_t)(idx >> 3) & 63u, f = (uint32_t)(idx >> 9) & 3u;
    uint64_t rest = idx >> 11;
    const uint32_t s = (uint32_t)(rest % kpg);
    rest /= kpg;
    const uint32_t g = (uint32_t)(rest % G), ct = (uint32_t)(rest / G);
    const uint32_t c = lane & 31u, h = lane >> 5, j = e >> 1, half = e & 1u;
    uint32_t word, mode;
    ew_kstep(kpg, s, word, mode);
    uint32_t pos;
    if (mode < 2u) {
        pos = ((half ? 30u : 14u) - j + 8u * mode + 4u * h) & 31u;
    } else {
        const uint32_t pp = ((half ? 30u : 14u) - j + 21u + 4u * h) & 31u;  // 0..7 or 16..23 of the compressed word
        pos = pp < 8u ? pp : pp - 8u;
    }
    const uint32_t row = 2u * g + (word >> 1), col = 32u * (word & 1u) + pos;
    const uint32_t n = ct * EW_COLS + 64u * (f >> 1) + 2u * c + (f & 1u);
    float v = 0.0f;
    if (col < cols) {
        v = (float)w[(uint64_t)n * ld + (uint64_t)row * cols + col];
        v *= __uint_as_float((127u + 128u - (128u >> j)) << 23);  // see pack_embed_kernel
        const float lim = 3.3895313892515355e38f;
        v = v > lim ? lim : v < -lim ? -lim : v;
    }
    out[idx] = __float2bfloat16(v);
}

struct EmbedWordsArgs {
    const uint4 *words;     // [B][groups] 16-byte groups = [B][rows] uint64
    const uint4 *wp;        // packed weights
    const float *bias;      // [hidden] f32 or null
    uint32_t *out;          // [B][ld_out / 2] bf16 pairs
    uint64_t B;
    uint64_t ld_out;        // elements per env row of out
    uint32_t groups;        // rows / 2
    uint32_t n_ctiles;      // hidden / 128
    uint32_t relu;
};

// NB = B fragments per k-step the workgroup computes: 4 (the packed column tile of 128) or 2 (one 64-column half of it: twice the
// workgroups, for batches that would otherwise leave CUs idle; the packed weights are the same, the half's fragments are picked while staging).
template <uint32_t KPG, uint32_t NB>
__global__ __launch_bounds__(64 * EW_WAVES, NB == 2 ? 3 : 2) void embed_words_kernel(EmbedWordsArgs a) {
    constexpr uint32_t CHUNK_VEC = KPG * NB * 64u;  // uint4 per streamed chunk
    __shared__ uint4 cbuf[2 * CHUNK_VEC];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_tiles_c = a.n_ctiles * (EW_NB / NB);  // column tiles of 32 NB
    const uint32_t ct = blockIdx.x % n_tiles_c;
    const uint64_t etile = blockIdx.x / n_tiles_c;
    const uint32_t G = a.groups;
    const uint32_t first_frag = NB == EW_NB ? 0u : (ct % (EW_NB / NB)) * NB;  // this workgroup's fragments inside the packed k-step
    const uint4 *wsrc = a.wp + (uint64_t)(ct / (EW_NB / NB)) * G * (KPG * EW_NB * 64u);
    auto stage = [&](uint32_t g) {  // chunk g -> buffer g & 1, one fragment (1 KiB) per wave instruction
        const uint4 *src = wsrc + (uint64_t)g * (KPG * EW_NB * 64u);
        uint4 *dst = cbuf + (g & 1u) * CHUNK_VEC;
#pragma unroll
        for (uint32_t j0 = 0; j0 < KPG * NB; j0 += EW_WAVES) {
            const uint32_t j = j0 + wave, ks = j / NB, f = j % NB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (ks * EW_NB + first_frag + f) * 64u + lane),
                                             (__attribute__((address_space(3))) void *)(dst + j * 64u), 16, 0, 0);
        }
    };
    auto landed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t shA = 4u * h, shB = 8u + 4u * h, shS = 21u + 4u * h;
    const uint64_t env0 = etile * EW_BLOCK_ENVS + (uint64_t)wave * (32u * EW_MA);
    const uint4 *pw[EW_MA];
#pragma unroll
    for (uint32_t i = 0; i < EW_MA; ++i) {
        uint64_t env = env0 + 32u * i + r;
        env = env < a.B ? env : a.B - 1;  // waves / rows past the batch compute on the last env (every wave takes part in the barriers)
        pw[i] = a.words + env * G;
    }
    f32x16 acc[EW_MA][NB];
#pragma unroll
    for (uint32_t i = 0; i < EW_MA; ++i)
#pragma unroll
        for (uint32_t f = 0; f < NB; ++f)
#pragma unroll
            for (uint32_t q = 0; q < 16; ++q) acc[i][f][q] = 0.0f;
    stage(0);
    uint4 cur[EW_MA], nxt[EW_MA];
#pragma unroll
    for (uint32_t i = 0; i < EW_MA; ++i) cur[i] = pw[i][0];
    landed();
    // A fragments of k-step s of the group held in `wds`
    auto expand = [&](const uint4 (&wds)[EW_MA], uint32_t s, bf16x8 (&af)[EW_MA]) {
        uint32_t word, mode;
        ew_kstep(KPG, s, word, mode);
#pragma unroll
        for (uint32_t i = 0; i < EW_MA; ++i) {
            uint32_t wv = word == 0 ? wds[i].x : word == 1 ? wds[i].y : word == 2 ? wds[i].z : wds[i].w;
            if (mode == 2u) wv = (wv & 0xFFu) | ((wv << 8) & 0xFF0000u);  // bits 8..15 -> 16..23
            af[i] = emb_expand(wv, mode == 0u ? shA : mode == 1u ? shB : shS);
        }
    };
    // Software pipeline inside a chunk: while the 8 MFMAs of k-step s issue, the 4 B fragments of s + 1 are read from LDS and its
    // A fragments expanded into the other register set.  The chunk's first k-step is fetched right after the barrier.
    bf16x8 af[2][EW_MA], bfr[2][NB];
    {
        const uint4 *bl = cbuf + lane;
#pragma unroll
        for (uint32_t f = 0; f < NB; ++f) bfr[0][f] = __builtin_bit_cast(bf16x8, bl[f * 64u]);
        expand(cur, 0, af[0]);
    }
    for (uint32_t g = 0; g < G; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1u < G) stage(g + 1u);  // its buffer was last read in chunk g - 1, which every wave has left
        const uint32_t gn = g + 1u < G ? g + 1u : g;
#pragma unroll
        for (uint32_t i = 0; i < EW_MA; ++i) nxt[i] = pw[i][gn];
        const uint4 *bl = cbuf + (g & 1u) * CHUNK_VEC + lane;
#pragma unroll
        for (uint32_t s = 0; s < KPG; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t c = s & 1u, n = c ^ 1u;
            if (s + 1u < KPG) {
#pragma unroll
                for (uint32_t f = 0; f < NB; ++f) bfr[n][f] = __builtin_bit_cast(bf16x8, bl[((s + 1u) * NB + f) * 64u]);
                expand(cur, s + 1u, af[n]);
            }
#pragma unroll
            for (uint32_t f = 0; f < NB; ++f)
#pragma unroll
                for (uint32_t i = 0; i < EW_MA; ++i) acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c][i], bfr[c][f], acc[i][f], 0, 0, 0);
            if (s + 1u < KPG) {
                // the first MFMA goes ahead of the next k-step's LDS reads: the wait for this k-step's fragments (read a whole k-step ago)
                // then finds no younger LDS read outstanding -- with the reads first the compiler's lgkmcnt(0) also waits for those
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
#pragma unroll
                for (uint32_t m = 1; m < NB * EW_MA; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x002, NB == 2 ? 4 : 2, 0);  // a share of the next k-step's expansion
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (uint32_t i = 0; i < EW_MA; ++i) cur[i] = nxt[i];
        landed();
        {   // KPG is even: the next chunk starts in register set 0 again
            const uint4 *bn = cbuf + ((g + 1u) & 1u) * CHUNK_VEC + lane;
#pragma unroll
            for (uint32_t f = 0; f < NB; ++f) bfr[0][f] = __builtin_bit_cast(bf16x8, bn[f * 64u]);
            expand(cur, 0, af[0]);
        }
    }
    // epilogue as in embed_bits_kernel: x 0.5 + bias, ReLU, bf16, 4x4 transpose over the lane quad, 16-byte stores
    const uint64_t ldw = a.ld_out >> 1;
    const uint32_t lane_row = (lane & 3u) + 4u * h;
#pragma unroll
    for (uint32_t sl = 0; sl < NB / 2u; ++sl) {
        const uint32_t n0 = ct * (32u * NB) + 64u * sl + 2u * r;
        const float bias0 = a.bias ? a.bias[n0] : 0.0f, bias1 = a.bias ? a.bias[n0 + 1] : 0.0f;
        uint32_t *const out_lane = a.out + ((ct * (32u * NB) + 64u * sl + 8u * (r >> 2)) >> 1);
#pragma unroll
        for (uint32_t i = 0; i < EW_MA; ++i) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                uint32_t d[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    float v0 = __builtin_fmaf(acc[i][2u * sl][4u * j + k], 0.5f, bias0), v1 = __builtin_fmaf(acc[i][2u * sl + 1u][4u * j + k], 0.5f, bias1);
                    if (a.relu) {
                        v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, __builtin_inff());
                        v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, __builtin_inff());
                    }
                    const f32x2 v = {v0, v1};
                    d[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                }
                emb_quad_swap<0xB1>(d[0], d[1], (lane & 1u) != 0);
                emb_quad_swap<0xB1>(d[2], d[3], (lane & 1u) != 0);
                emb_quad_swap<0x4E>(d[0], d[2], (lane & 2u) != 0);
                emb_quad_swap<0x4E>(d[1], d[3], (lane & 2u) != 0);
                const uint64_t row = env0 + 32u * i + 8u * j + lane_row;
                if (row < a.B) *reinterpret_cast<uint4 *>(out_lane + row * ldw) = make_uint4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

// Small batches: embed_words_kernel's workgroup takes 256 (NB = 4) or 128 envs under a 128- or 64-column tile whose weights stream
// through LDS, so 1 024 PauliGym envs occupy a few dozen CUs for 16 us.  As in embed_small_kernel a wave here owns one tile of 32 envs
// and ONE of the four fragment columns of a 128-column tile, and takes its fragments (1 KiB each, the packed layout is fragment order)
// from L2 straight into registers, EWS_AHEAD 16-byte groups ahead; the two waves of a workgroup share the env tile (and a CU) and take
// fragments f and f + 1 of a column-tile half.  Same packed weights, same expansion, same k order: bit-identical activations.
constexpr uint32_t EWS_AHEAD = 4;  // groups (KPG fragments + one 16-byte word group each) in flight per wave

template <uint32_t KPG>
__global__ __launch_bounds__(128) void embed_words_small_kernel(EmbedWordsArgs a) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per_tile = a.n_ctiles * 2u;  // workgroups per env tile: (column tile, half)
    const uint32_t ch = blockIdx.x % per_tile, ct = ch >> 1, f = (ch & 1u) * 2u + wave;  // this wave's fragment of every k-step
    const uint64_t tile = blockIdx.x / per_tile;
    const uint32_t G = a.groups;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t shA = 4u * h, shB = 8u + 4u * h, shS = 21u + 4u * h;
    uint64_t env = tile * 32u + r;
    env = env < a.B ? env : a.B - 1;  // rows past the batch compute on the last env and are not stored
    const uint4 *pw = a.words + env * G;
    const uint4 *wf = a.wp + (uint64_t)ct * G * (KPG * EW_NB * 64u) + f * 64u + lane;  // fragment (group g, k-step s) = wf[(g * KPG + s) * EW_NB * 64]
    uint4 wds[EWS_AHEAD], wb[EWS_AHEAD][KPG];
    auto fetch = [&](uint32_t g, uint32_t slot) {
        wds[slot] = pw[g];
#pragma unroll
        for (uint32_t s = 0; s < KPG; ++s) wb[slot][s] = wf[(uint64_t)(g * KPG + s) * (EW_NB * 64u)];
    };
    f32x16 acc;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) acc[q] = 0.0f;
    auto multiply = [&](uint32_t slot) {
#pragma unroll
        for (uint32_t s = 0; s < KPG; ++s) {
            uint32_t word, mode;
            ew_kstep(KPG, s, word, mode);
            uint32_t wv = word == 0 ? wds[slot].x : word == 1 ? wds[slot].y : word == 2 ? wds[slot].z : wds[slot].w;
            if (mode == 2u) wv = (wv & 0xFFu) | ((wv << 8) & 0xFF0000u);  // bits 8..15 -> 16..23 (embed_words_kernel)
            const bf16x8 af = emb_expand(wv, mode == 0u ? shA : mode == 1u ? shB : shS);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, wb[slot][s]), acc, 0, 0, 0);
        }
    };
#pragma unroll
    for (uint32_t g = 0; g + 1u < EWS_AHEAD; ++g)
        if (g < G) fetch(g, g);
    for (uint32_t g0 = 0; g0 < G; g0 += EWS_AHEAD) {  // EWS_AHEAD groups per trip: the register ring is indexed statically
#pragma unroll
        for (uint32_t k = 0; k < EWS_AHEAD; ++k) {
            const uint32_t g = g0 + k;
            if (g < G) {
                if (g + EWS_AHEAD - 1u < G) fetch(g + EWS_AHEAD - 1u, (k + EWS_AHEAD - 1u) % EWS_AHEAD);
                multiply(k);
            }
        }
    }
    // x 0.5 + bias, ReLU, bf16; C layout: column ct * 128 + 64 (f >> 1) + 2 (lane & 31) + (f & 1), env row = (q & 3) + 8 (q >> 2) + 4 h
    const uint32_t n0 = ct * EW_COLS + 64u * (f >> 1) + 2u * r + (f & 1u);
    const float bias0 = a.bias ? a.bias[n0] : 0.0f;
    __hip_bfloat16 *out = reinterpret_cast<__hip_bfloat16 *>(a.out) + n0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
        float v0 = __builtin_fmaf(acc[q], 0.5f, bias0);
        if (a.relu) v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, __builtin_inff());
        const uint64_t e = tile * 32u + (q & 3u) + 8u * (q >> 2) + 4u * h;
        if (e < a.B) out[e * a.ld_out] = __float2bfloat16(v0);
    }
}

}  // namespace qg

using namespace qg;

extern "C" {

// Layouts whose resident rows are uint32 words in 16-byte groups of four, tiles of 64 envs: TILE, and LFD with N <= 32 (two regions per tile)
static bool emb_layout(const qg_vec *v) { return v->layout == LAYOUT_TILE || (v->layout == LAYOUT_LFD && !v->w64); }
static uint32_t emb_rows(const qg_vec *v) { return v->layout == LAYOUT_LFD ? 4u * v->nxp : (v->has_z ? 2 * v->nxp : v->nxp); }  // row slots per matrix
#define QG_EMB_LAYOUTS "the bit-consuming first layer needs uint32 rows resident in tiles (CliffordEnv N <= 16, LinearFunctionEnv 8 < N <= 32)"

size_t qg_vec_embed_packed_bytes(const qg_vec *v, uint32_t hidden) {
    if (!v || !emb_layout(v) || hidden == 0 || hidden % EMB_SLAB) return 0;
    return (size_t)hidden * 16u * emb_ksteps(emb_rows(v)) * 2u;
}

int qg_vec_pack_embedding(qg_vec *v, const void *weight_dev, int weight_dtype, uint64_t ld, uint32_t hidden, void *packed_dev, void *stream) {
    if (!v || !weight_dev || !packed_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (!emb_layout(v)) return set_error(QG_ERR_UNSUPPORTED, QG_EMB_LAYOUTS);
    if (hidden == 0 || hidden % EMB_SLAB) return set_error(QG_ERR_INVALID, "hidden size must be a multiple of %u", EMB_SLAB);
    if (ld < (uint64_t)v->D * v->D) return set_error(QG_ERR_INVALID, "weight rows are shorter than the observation (%u x %u)", v->D, v->D);
    QG_ON_DEVICE(v);
    const uint32_t R = emb_rows(v);
    if (!v->embed_dump) HIP_TRY(hipMalloc(&v->embed_dump, 1024));  // see EmbedArgs::dump
    const uint64_t total = (uint64_t)hidden * 16u * emb_ksteps(R);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    __hip_bfloat16 *out = reinterpret_cast<__hip_bfloat16 *>(packed_dev);
    switch (weight_dtype) {
    case QG_DT_F32:
        hipLaunchKernelGGL(pack_embed_kernel<float>, grid, block, 0, s, reinterpret_cast<const float *>(weight_dev), ld, hidden, R, v->N, v->D, (uint32_t)v->has_z, out);
        break;
    case QG_DT_BF16:
        hipLaunchKernelGGL(pack_embed_kernel<__hip_bfloat16>, grid, block, 0, s, reinterpret_cast<const __hip_bfloat16 *>(weight_dev), ld, hidden, R, v->N, v->D,
                           (uint32_t)v->has_z, out);
        break;
    default: return set_error(QG_ERR_INVALID, "weight dtype must be f32 or bf16");
    }
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

static int embed_impl(qg_vec *v, const void *packed_dev, const float *bias_dev, uint32_t hidden, int relu, void *out_dev, uint64_t ld_out, void *obs_dev,
                      void *stream) {
    if (!v || !packed_dev || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (!emb_layout(v)) return set_error(QG_ERR_UNSUPPORTED, QG_EMB_LAYOUTS);
    if (hidden == 0 || hidden % EMB_SLAB) return set_error(QG_ERR_INVALID, "hidden size must be a multiple of %u", EMB_SLAB);
    if (ld_out < hidden || (ld_out & 7u) || (reinterpret_cast<uintptr_t>(out_dev) & 15u))
        return set_error(QG_ERR_INVALID, "the output must be 16-byte aligned with a row stride that is a multiple of 8 elements");
    QG_ON_DEVICE(v);
    const uint32_t R = emb_rows(v);
    EmbedArgs a;
    a.state = reinterpret_cast<const uint4 *>(v->state);
    a.wp = reinterpret_cast<const uint4 *>(packed_dev);
    a.bias = bias_dev;
    a.out = reinterpret_cast<uint32_t *>(out_dev);
    if (!v->embed_dump) return set_error(QG_ERR_INVALID, "qg_vec_pack_embedding must run on this handle first");
    a.dump = reinterpret_cast<uint4 *>(v->embed_dump);
    a.B = v->B;
    a.ld_out = ld_out;
    a.n_slabs = hidden / EMB_SLAB;
    a.relu = relu ? 1u : 0u;
    a.obs = nullptr;
    a.N = v->N;
    a.D = v->has_z ? 2u * v->N : v->N;
    a.has_z = v->has_z ? 1u : 0u;
    const uint32_t G = R / 4;
    a.region = v->layout == LAYOUT_LFD ? v->inverted : nullptr;
    a.tile_groups = v->layout == LAYOUT_LFD ? 2u * G : G;
    const size_t lds = (size_t)emb_groups(R) * 8u * 2u * 64u * 16u;  // <= 128 KiB (R <= 32)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, v->device);
    hipStream_t s = (hipStream_t)stream;
    const uint64_t env_tiles = (v->B + 31u) / 32u;
    // (32-env tile, slab) workgroups of embed_small_kernel: too few envs for 512-env passes under an LDS-resident slab.  Measured inside the
    // CliffordGym collector: 21 against 23 us per step at 2 048 envs (two workgroups per CU), equal at 4 096 (four), slower beyond.
    const bool small = env_tiles * a.n_slabs <= (uint64_t)EMS_WGS_PER_CU * (uint64_t)cus;
    if (obs_dev) {
        if (small && (reinterpret_cast<uintptr_t>(obs_dev) & 15u) == 0) a.obs = reinterpret_cast<uint32_t *>(obs_dev);  // written by the same launch
        else if (const int rc = qg_vec_observe_packed(v, obs_dev, stream)) return rc;
    }
    if (small) {
        const dim3 grid((unsigned)(env_tiles * a.n_slabs)), block(128);
#define QG_EMS_CASE(GG)                                                           \
    case GG: hipLaunchKernelGGL(embed_small_kernel<GG>, grid, block, 0, s, a); break;
        switch (G) {
            QG_EMS_CASE(2) QG_EMS_CASE(3) QG_EMS_CASE(4) QG_EMS_CASE(5) QG_EMS_CASE(6) QG_EMS_CASE(7) QG_EMS_CASE(8)
        default: return set_error(QG_ERR_UNSUPPORTED, "unexpected row-group count %u", G);
        }
#undef QG_EMS_CASE
        HIP_TRY(hipGetLastError());
        return QG_OK;
    }
    const uint64_t env_blocks = (v->B + EMB_BLOCK_ENVS - 1) / EMB_BLOCK_ENVS;
    uint32_t mgroups = (uint32_t)((uint32_t)cus / a.n_slabs);
    if (mgroups == 0) mgroups = 1;
    if (mgroups > env_blocks) mgroups = (uint32_t)env_blocks;
    const dim3 grid(mgroups * a.n_slabs), block(EMB_THREADS);
#define QG_EMB_CASE(GG)                                                                                                       \
    case GG:                                                                                                                  \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(embed_bits_kernel<GG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(embed_bits_kernel<GG>, grid, block, lds, s, a);                                                     \
        break;
    switch (G) {
        QG_EMB_CASE(2) QG_EMB_CASE(3) QG_EMB_CASE(4) QG_EMB_CASE(5) QG_EMB_CASE(6) QG_EMB_CASE(7) QG_EMB_CASE(8)
    default: return set_error(QG_ERR_UNSUPPORTED, "unexpected row-group count %u", G);
    }
#undef QG_EMB_CASE
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

int qg_vec_embed(qg_vec *v, const void *packed_dev, const float *bias_dev, uint32_t hidden, int relu, void *out_dev, uint64_t ld_out, void *stream) {
    return embed_impl(v, packed_dev, bias_dev, hidden, relu, out_dev, ld_out, nullptr, stream);
}

int qg_vec_embed_observe(qg_vec *v, const void *packed_dev, const float *bias_dev, uint32_t hidden, int relu, void *out_dev, uint64_t ld_out,
                         void *obs_packed_dev, void *stream) {
    if (!obs_packed_dev) return set_error(QG_ERR_INVALID, "null argument");
    return embed_impl(v, packed_dev, bias_dev, hidden, relu, out_dev, ld_out, obs_packed_dev, stream);
}

size_t qg_policy_embed_words_packed_bytes(uint32_t rows, uint32_t cols, uint32_t hidden) {
    if (rows == 0 || (rows & 1u) || cols == 0 || cols > 64u || hidden == 0 || hidden % EW_COLS) return 0;
    return (size_t)(hidden / EW_COLS) * (rows / 2u) * ew_kpg(cols) * EW_NB * 64u * 16u;
}

int qg_policy_pack_embed_words(const void *weight_dev, int weight_dtype, uint64_t ld, uint32_t rows, uint32_t cols, uint32_t hidden, void *packed_dev,
                               void *stream) {
    if (!weight_dev || !packed_dev) return set_error(QG_ERR_INVALID, "null argument");
    const size_t bytes = qg_policy_embed_words_packed_bytes(rows, cols, hidden);
    if (bytes == 0) return set_error(QG_ERR_UNSUPPORTED, "first layer from packed words: an even number of rows, <= 64 columns, hidden a multiple of %u", EW_COLS);
    if (ld < (uint64_t)rows * cols) return set_error(QG_ERR_INVALID, "weight rows are shorter than the observation (%u x %u)", rows, cols);
    const uint64_t total = bytes / 2u;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    __hip_bfloat16 *out = reinterpret_cast<__hip_bfloat16 *>(packed_dev);
    switch (weight_dtype) {
    case QG_DT_F32:
        hipLaunchKernelGGL(pack_embed_words_kernel<float>, grid, block, 0, s, reinterpret_cast<const float *>(weight_dev), ld, hidden, rows, cols, out);
        break;
    case QG_DT_BF16:
        hipLaunchKernelGGL(pack_embed_words_kernel<__hip_bfloat16>, grid, block, 0, s, reinterpret_cast<const __hip_bfloat16 *>(weight_dev), ld, hidden, rows, cols,
                           out);
        break;
    default: return set_error(QG_ERR_INVALID, "weight dtype must be f32 or bf16");
    }
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

int qg_policy_embed_words(const uint64_t *words_dev, uint64_t batch, uint32_t rows, uint32_t cols, const void *packed_dev, const float *bias_dev,
                          uint32_t hidden, int relu, void *out_dev, uint64_t ld_out, void *stream) {
    if (!words_dev || !packed_dev || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (qg_policy_embed_words_packed_bytes(rows, cols, hidden) == 0)
        return set_error(QG_ERR_UNSUPPORTED, "first layer from packed words: an even number of rows, <= 64 columns, hidden a multiple of %u", EW_COLS);
    if ((reinterpret_cast<uintptr_t>(words_dev) & 15u)) return set_error(QG_ERR_INVALID, "the packed words must be 16-byte aligned");
    if (ld_out < hidden || (ld_out & 7u) || (reinterpret_cast<uintptr_t>(out_dev) & 15u))
        return set_error(QG_ERR_INVALID, "the output must be 16-byte aligned with a row stride that is a multiple of 8 elements");
    if (batch == 0) return QG_OK;
    EmbedWordsArgs a;
    a.words = reinterpret_cast<const uint4 *>(words_dev);
    a.wp = reinterpret_cast<const uint4 *>(packed_dev);
    a.bias = bias_dev;
    a.out = reinterpret_cast<uint32_t *>(out_dev);
    a.B = batch;
    a.ld_out = ld_out;
    a.groups = rows / 2u;
    a.n_ctiles = hidden / EW_COLS;
    a.relu = relu ? 1u : 0u;
    const uint64_t etiles = (batch + EW_BLOCK_ENVS - 1) / EW_BLOCK_ENVS;
    if (etiles * a.n_ctiles > 0x7FFFFFFFull) return set_error(QG_ERR_UNSUPPORTED, "batch too large for one launch");
    const dim3 grid((unsigned)(etiles * a.n_ctiles)), block(64 * EW_WAVES);
    hipStream_t s = (hipStream_t)stream;
    // 64-column workgroup tiles while 128-column ones would not give every CU its two workgroups
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t small_wgs = ((batch + 31u) / 32u) * a.n_ctiles * 2u;
    if (small_wgs <= 2ull * (uint64_t)cus) {  // at most two workgroups of embed_words_small_kernel per CU (PauliGym 20q collector: 36 against 41 us per step at 1 024 envs, 39 against 42 at 2 048; 46 against 45 at 4 096)
        const dim3 sgrid((unsigned)small_wgs), sblock(128);
        switch (ew_kpg(cols)) {
        case 4: hipLaunchKernelGGL(embed_words_small_kernel<4>, sgrid, sblock, 0, s, a); break;
        case 6: hipLaunchKernelGGL(embed_words_small_kernel<6>, sgrid, sblock, 0, s, a); break;
        default: hipLaunchKernelGGL(embed_words_small_kernel<8>, sgrid, sblock, 0, s, a); break;
        }
        HIP_TRY(hipGetLastError());
        return QG_OK;
    }
    const bool narrow = etiles * a.n_ctiles < 2ull * (uint64_t)cus;
    const dim3 ngrid((unsigned)(etiles * a.n_ctiles * 2u));
    switch (ew_kpg(cols) * 10u + (narrow ? 2u : 4u)) {
    case 42: hipLaunchKernelGGL((embed_words_kernel<4, 2>), ngrid, block, 0, s, a); break;
    case 44: hipLaunchKernelGGL((embed_words_kernel<4, 4>), grid, block, 0, s, a); break;
    case 62: hipLaunchKernelGGL((embed_words_kernel<6, 2>), ngrid, block, 0, s, a); break;
    case 64: hipLaunchKernelGGL((embed_words_kernel<6, 4>), grid, block, 0, s, a); break;
    case 82: hipLaunchKernelGGL((embed_words_kernel<8, 2>), ngrid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((embed_words_kernel<8, 4>), grid, block, 0, s, a); break;
    }
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

size_t qg_policy_head_packed_bytes(uint32_t num_actions, uint32_t in_features) {
    const uint32_t tiles = (num_actions + 1u + 31u) / 32u;
    if (num_actions == 0 || tiles > 7 || in_features == 0 || in_features % 64u || in_features > 512u) return 0;
    const size_t bytes = (size_t)(in_features / 16u + 1u) * tiles * 64u * 16u;
    if (bytes > 144u * 1024u) return 0;
    return (bytes + 16383u) / 16384u * 16384u;  // whole 16 KiB chunks: qg_policy_mid_head_sample streams the head through LDS in such pieces
}

static int pack_rows_impl(const void *weight_dev, const void *bias_dev, int dtype, uint64_t ld, uint32_t in_features, uint32_t rows,
                          int32_t value_row, uint32_t tiles, uint32_t xorder, uint32_t ks_total, void *packed_dev, hipStream_t s) {
    const uint64_t total = (uint64_t)ks_total * tiles * 64u * 8u;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    __hip_bfloat16 *out = reinterpret_cast<__hip_bfloat16 *>(packed_dev);
    switch (dtype) {
    case QG_DT_F32:
        hipLaunchKernelGGL(pack_head_kernel<float>, grid, block, 0, s, reinterpret_cast<const float *>(weight_dev), reinterpret_cast<const float *>(bias_dev), ld,
                           in_features, rows, value_row, tiles, xorder, ks_total, out);
        break;
    case QG_DT_BF16:
        hipLaunchKernelGGL(pack_head_kernel<__hip_bfloat16>, grid, block, 0, s, reinterpret_cast<const __hip_bfloat16 *>(weight_dev),
                           reinterpret_cast<const __hip_bfloat16 *>(bias_dev), ld, in_features, rows, value_row, tiles, xorder, ks_total, out);
        break;
    default: return set_error(QG_ERR_INVALID, "weight dtype must be f32 or bf16");
    }
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

int qg_policy_pack_head(const void *weight_dev, const void *bias_dev, int dtype, uint64_t ld, uint32_t in_features, uint32_t num_actions,
                        int32_t value_row, int after_mid, void *packed_dev, void *stream) {
    if (!weight_dev || !packed_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (qg_policy_head_packed_bytes(num_actions, in_features) == 0)
        return set_error(QG_ERR_UNSUPPORTED, "fused head: num_actions <= 222, in_features a multiple of 64 and <= 512, packed head <= 144 KiB");
    if (ld < in_features) return set_error(QG_ERR_INVALID, "weight rows are shorter than in_features");
    const uint32_t tiles = (num_actions + 1u + 31u) / 32u;
    return pack_rows_impl(weight_dev, bias_dev, dtype, ld, in_features, num_actions, value_row, tiles, after_mid ? 1u : 0u, in_features / 16u + 1u, packed_dev,
                          (hipStream_t)stream);
}

size_t qg_policy_mid_packed_bytes(uint32_t in_features, uint32_t mid_features) {
    if (mid_features != 32u * MID_FT || in_features == 0 || in_features % (16u * MID_CHUNK) || in_features > 2048u) return 0;
    return (size_t)(in_features / 16u + MID_CHUNK) * MID_FT * 64u * 16u;
}

int qg_policy_pack_mid(const void *weight_dev, const void *bias_dev, int dtype, uint64_t ld, uint32_t in_features, uint32_t mid_features,
                       void *packed_dev, void *stream) {
    if (!weight_dev || !packed_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (qg_policy_mid_packed_bytes(in_features, mid_features) == 0)
        return set_error(QG_ERR_UNSUPPORTED, "fused middle layer: 256 output features, in_features a multiple of 32 and <= 2048");
    if (ld < in_features) return set_error(QG_ERR_INVALID, "weight rows are shorter than in_features");
    return pack_rows_impl(weight_dev, bias_dev, dtype, ld, in_features, mid_features, -1, MID_FT, 0u, in_features / 16u + MID_CHUNK, packed_dev, (hipStream_t)stream);
}

int qg_policy_head_sample(const void *h_dev, uint64_t ld_h, uint64_t batch, uint32_t in_features, const void *packed_dev, uint32_t num_actions,
                          uint64_t seed, uint64_t counter, const uint64_t *clock_dev, void *actions_dev, int action_dtype, float *logp_dev,
                          float *entropy_dev, float *values_dev, void *stream) {
    if (!h_dev || !packed_dev || !actions_dev) return set_error(QG_ERR_INVALID, "null argument");
    const size_t lds = qg_policy_head_packed_bytes(num_actions, in_features);
    if (lds == 0) return set_error(QG_ERR_UNSUPPORTED, "fused head: num_actions <= 222, in_features a multiple of 64 and <= 512, packed head <= 144 KiB");
    if (ld_h < in_features || (ld_h & 7u) || (reinterpret_cast<uintptr_t>(h_dev) & 15u))
        return set_error(QG_ERR_INVALID, "activations must be 16-byte aligned bf16 rows with a stride that is a multiple of 8");
    if (action_dtype != QG_ACT_I32 && action_dtype != QG_ACT_I64) return set_error(QG_ERR_INVALID, "bad action dtype");
    if (batch == 0) return QG_OK;
    HeadArgs a;
    a.h = reinterpret_cast<const uint4 *>(h_dev);
    a.wp = reinterpret_cast<const uint4 *>(packed_dev);
    a.actions = actions_dev;
    a.logp = logp_dev;
    a.entropy = entropy_dev;
    a.values = values_dev;
    a.clock = clock_dev;
    a.ld_h = ld_h;
    a.B = batch;
    a.seed = seed ^ 0x73616D70ull;  // the stream of qg_sample_actions
    a.counter = counter;
    a.K = in_features;
    a.A = num_actions;
    a.act64 = action_dtype == QG_ACT_I64;
    a.env_base = 0;
    const uint32_t tiles = (num_actions + 1u + 31u) / 32u;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t env_tiles = (batch + 31u) / 32u, want = (env_tiles + HEAD_WAVES - 1) / HEAD_WAVES;
    const dim3 grid((unsigned)(want < (uint64_t)cus ? want : (uint64_t)cus)), block(64 * HEAD_WAVES);
    hipStream_t s = (hipStream_t)stream;
#define QG_HEAD_CASE(TT)                                                                                                      \
    case TT:                                                                                                                  \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(head_sample_kernel<TT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(head_sample_kernel<TT>, grid, block, lds, s, a);                                                   \
        break;
    switch (tiles) {
        QG_HEAD_CASE(1) QG_HEAD_CASE(2) QG_HEAD_CASE(3) QG_HEAD_CASE(4) QG_HEAD_CASE(5) QG_HEAD_CASE(6) QG_HEAD_CASE(7)
    default: return set_error(QG_ERR_UNSUPPORTED, "too many actions for the fused head");
    }
#undef QG_HEAD_CASE
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

// up to one workgroup per CU of mid_head_small_kernel (a tile of 32 envs each)
static bool mid_head_is_small(uint64_t batch, uint32_t in_features, int cus) {
    return (batch + 31u) / 32u <= (uint64_t)cus && in_features % (16u * MHS_GROUP) == 0 && in_features <= 16u * MHS_KMAX;
}

// mid_head_sample_kernel with two env tiles per wave (one 256-env workgroup per CU): built and measured in round 4 -- bit-identical
// (tests/test_gpu_collect_ops.py, test_gpu_collector.py) and 2.5x SLOWER (92.9 against 36.5 us at 65 536 envs; collector 159 against 104 us
// per step): at one wave per SIMD nothing covers a chunk's memory round trip (every chunk ends in s_waitcnt vmcnt(0) + barrier with the
// activation prefetch and the next chunk's staging in flight, and the 32 MFMAs of a chunk are shorter than that trip), and the two tiles'
// accumulators (256 AGPRs + 256 VGPRs) leave the draw 266 spilled registers (profiles/r04/mid_head_two_tiles.txt).  Kept behind a
// development build flag (tools/build_variant.sh two_tiles kernels_policy.hip -DQG_MID_HEAD_TWO_TILES); the product uses one tile per wave.
static bool mid_head_two_tiles(uint64_t batch, int cus) {
#ifdef QG_MID_HEAD_TWO_TILES
    return batch >= 256ull * (uint64_t)cus;
#else
    (void)batch;
    (void)cus;
    return false;
#endif
}

static int mid_head_impl(const void *h_dev, uint64_t ld_h, uint64_t batch, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                         const void *packed_head_dev, uint32_t num_actions, uint64_t seed, uint64_t counter, const uint64_t *clock_dev, void *actions_dev,
                         int action_dtype, float *logp_dev, float *entropy_dev, float *values_dev, qg_vec *step_of, float *step_rewards,
                         uint8_t *step_dones, const uint64_t *reset_seed, uint64_t env_base, void *stream) {
    if (!h_dev || !packed_mid_dev || !packed_head_dev || !actions_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (qg_policy_head_packed_bytes(num_actions, mid_features) == 0 || qg_policy_mid_packed_bytes(in_features, mid_features) == 0)
        return set_error(QG_ERR_UNSUPPORTED, "fused middle layer + head: 256 middle features, in_features a multiple of 32, num_actions <= 222");
    if (ld_h < in_features || (ld_h & 7u) || (reinterpret_cast<uintptr_t>(h_dev) & 15u))
        return set_error(QG_ERR_INVALID, "activations must be 16-byte aligned bf16 rows with a stride that is a multiple of 8");
    if (action_dtype != QG_ACT_I32 && action_dtype != QG_ACT_I64) return set_error(QG_ERR_INVALID, "bad action dtype");
    if (batch == 0) return QG_OK;
    MidHeadArgs m;
    HeadArgs &a = m.head;
    a.h = reinterpret_cast<const uint4 *>(h_dev);
    a.wp = reinterpret_cast<const uint4 *>(packed_head_dev);
    a.actions = actions_dev;
    a.logp = logp_dev;
    a.entropy = entropy_dev;
    a.values = values_dev;
    a.clock = clock_dev;
    a.ld_h = ld_h;
    a.B = batch;
    a.seed = seed ^ 0x73616D70ull;
    a.counter = counter;
    a.K = mid_features;
    a.A = num_actions;
    a.act64 = action_dtype == QG_ACT_I64;
    a.env_base = env_base;  // a shard draws what its envs would draw in the unsharded batch (qg_vec_set_env_base of the stepped handle)
    m.w2p = reinterpret_cast<const uint4 *>(packed_mid_dev);
    m.K1 = in_features;
    if (step_of) {  // Env::step with the drawn action in the same launch
        qg_vec *v = step_of;
        fill_step_args_public(v, m.step);
        m.step.rewards_seq = step_rewards;
        m.step.dones_seq = step_dones;
        m.step_groups = (v->has_z ? 2u * v->nxp : v->nxp) / 4u;
        m.step_has_z = v->has_z ? 1u : 0u;
        m.done_list = v->done_list;
        m.done_count = v->done_list + v->B;
    } else {
        m.step = StepArgs{};
        m.step_groups = m.step_has_z = 0;
        m.done_list = m.done_count = nullptr;
    }
    m.reset = InitArgs{};
    const uint32_t tiles = (num_actions + 1u + 31u) / 32u;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t env_tiles = (batch + 31u) / 32u, want = (env_tiles + MH_WAVES - 1) / MH_WAVES;
    hipStream_t s = (hipStream_t)stream;
    // up to one workgroup per CU of mid_head_small_kernel (a tile of 32 envs each): the batch is too small to fill the chip with
    // mid_head_sample_kernel's 128-env workgroups
    const bool small = mid_head_is_small(batch, in_features, cus);
    // the finished envs' indices are appended to the handle's list, unless the small kernel resets them itself
    const bool reset_in_kernel = small && step_of && reset_seed;
    bool trusted = false;
    if (step_of) trusted = done_list_session(step_of, s);
    if (reset_in_kernel) fill_reset_done_args_public(step_of, *reset_seed, m.reset);
    else if (step_of)  // the kernel appends the envs it finishes: the list's length is zero when it starts
        if (int rc = done_list_before_append(step_of, s)) return rc;
    if (small) {
        const dim3 grid((unsigned)env_tiles), block(64 * MHS_WAVES);
#define QG_MHS_CASE(TT)                                                        \
    case TT: hipLaunchKernelGGL(mid_head_small_kernel<TT>, grid, block, 0, s, m); break;
        switch (tiles) {
            QG_MHS_CASE(1) QG_MHS_CASE(2) QG_MHS_CASE(3) QG_MHS_CASE(4) QG_MHS_CASE(5) QG_MHS_CASE(6) QG_MHS_CASE(7)
        default: return set_error(QG_ERR_UNSUPPORTED, "too many actions for the fused head");
        }
#undef QG_MHS_CASE
        HIP_TRY(hipGetLastError());
        if (step_of && !reset_in_kernel) done_list_appended(step_of, trusted);
        return QG_OK;
    }
    // two tiles per wave (one workgroup of 256 envs per CU) once the batch gives every CU such a workgroup; below that, one tile per wave, two
    // workgroups per CU (32 KiB of LDS and 256 registers x 4 waves each)
    const bool two = mid_head_two_tiles(batch, cus);
    const uint64_t per_wg = two ? 2ull * MH_WAVES : (uint64_t)MH_WAVES, want_wg = (env_tiles + per_wg - 1) / per_wg;
    const uint64_t resident = two ? (uint64_t)cus : 2ull * (uint64_t)cus;
    const dim3 grid((unsigned)(want_wg < resident ? want_wg : resident)), block(64 * MH_WAVES);
#ifdef QG_MID_HEAD_TWO_TILES
#define QG_MH_CASE(TT)                                                                                   \
    case TT:                                                                                             \
        if (two) hipLaunchKernelGGL((mid_head_sample_kernel<TT, 2>), grid, block, 0, s, m);              \
        else hipLaunchKernelGGL((mid_head_sample_kernel<TT, 1>), grid, block, 0, s, m);                  \
        break;
#else
#define QG_MH_CASE(TT)                                                                                   \
    case TT: hipLaunchKernelGGL((mid_head_sample_kernel<TT, 1>), grid, block, 0, s, m); break;
#endif
    switch (tiles) {
        QG_MH_CASE(1) QG_MH_CASE(2) QG_MH_CASE(3) QG_MH_CASE(4) QG_MH_CASE(5) QG_MH_CASE(6) QG_MH_CASE(7)
    default: return set_error(QG_ERR_UNSUPPORTED, "too many actions for the fused head");
    }
#undef QG_MH_CASE
    HIP_TRY(hipGetLastError());
    if (step_of) done_list_appended(step_of, trusted);
    return QG_OK;
}

int qg_policy_mid_head_sample(const void *h_dev, uint64_t ld_h, uint64_t batch, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                              const void *packed_head_dev, uint32_t num_actions, uint64_t seed, uint64_t counter, const uint64_t *clock_dev, void *actions_dev,
                              int action_dtype, float *logp_dev, float *entropy_dev, float *values_dev, void *stream) {
    return mid_head_impl(h_dev, ld_h, batch, in_features, packed_mid_dev, mid_features, packed_head_dev, num_actions, seed, counter, clock_dev, actions_dev,
                         action_dtype, logp_dev, entropy_dev, values_dev, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

static int mid_head_step_impl(qg_vec *v, const void *h_dev, uint64_t ld_h, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                              const void *packed_head_dev, uint64_t seed, uint64_t counter, void *actions_dev, int action_dtype, float *logp_dev,
                              float *entropy_dev, float *values_dev, float *rewards_dev, uint8_t *dones_dev, const uint64_t *reset_seed, void *stream) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    const bool inverts = v->flags & F_INVERTS;
    if (v->layout != LAYOUT_TILE || (!inverts && !v->bad) || !v->done_list)
        return set_error(QG_ERR_UNSUPPORTED, "the sampling kernel steps TILE-layout handles (CliffordEnv N <= 16; LinearFunctionEnv 8 < N <= 32 without add_inverts)");
    if (reset_seed && v->gates.empty() && v->difficulty)  // Uniform::new(0, 0) panics in the reference
        return set_error(QG_ERR_PANIC, "reset with an empty gateset (the reference panics in Uniform::new(0, 0))");
    QG_ON_DEVICE(v);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, v->device);
    // add_inverts (CliffordEnv's default): the step is the two-lanes-per-env kernel's, which the small-batch sampling kernel can run on its
    // own lanes (qm_inv2_body) when every env is symplectic; otherwise the sampling kernel and the env's own step launch
    if (inverts && !(mid_head_is_small(v->B, in_features, cus) && v->has_z && !v->maybe_nonsymplectic)) {
        int rc = mid_head_impl(h_dev, ld_h, v->B, in_features, packed_mid_dev, mid_features, packed_head_dev, (uint32_t)v->gates.size(), seed, counter,
                               v->clock_dev, actions_dev, action_dtype, logp_dev, entropy_dev, values_dev, nullptr, nullptr, nullptr, nullptr, v->env_base, stream);
        if (rc == QG_OK) rc = qg_vec_rollout(v, actions_dev, action_dtype, 1, nullptr, rewards_dev, dones_dev, 0, stream);
        if (rc == QG_OK && reset_seed) rc = qg_vec_reset_done(v, *reset_seed, stream);
        return rc;
    }
    const int rc = mid_head_impl(h_dev, ld_h, v->B, in_features, packed_mid_dev, mid_features, packed_head_dev, (uint32_t)v->gates.size(), seed, counter,
                                 v->clock_dev, actions_dev, action_dtype, logp_dev, entropy_dev, values_dev, v, rewards_dev, dones_dev, reset_seed, v->env_base, stream);
    if (rc != QG_OK) return rc;
    v->step_index += 1;
    if (v->dense)  // qg_vec_track_dense: the sampling kernel's step does not write the dense observation (its policy reads the bits)
        if (int rc2 = dense_refresh_public(v, (hipStream_t)stream)) return rc2;
    // larger batches: the kernel left the list of finished envs, the reset is its own launch
    if (reset_seed && !mid_head_is_small(v->B, in_features, cus)) return qg_vec_reset_done(v, *reset_seed, stream);
    return QG_OK;
}

int qg_vec_mid_head_sample_step(qg_vec *v, const void *h_dev, uint64_t ld_h, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                                const void *packed_head_dev, uint64_t seed, uint64_t counter, void *actions_dev, int action_dtype, float *logp_dev,
                                float *entropy_dev, float *values_dev, float *rewards_dev, uint8_t *dones_dev, void *stream) {
    return mid_head_step_impl(v, h_dev, ld_h, in_features, packed_mid_dev, mid_features, packed_head_dev, seed, counter, actions_dev, action_dtype, logp_dev,
                              entropy_dev, values_dev, rewards_dev, dones_dev, nullptr, stream);
}

int qg_vec_mid_head_sample_step_reset(qg_vec *v, const void *h_dev, uint64_t ld_h, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                                      const void *packed_head_dev, uint64_t seed, uint64_t counter, void *actions_dev, int action_dtype, float *logp_dev,
                                      float *entropy_dev, float *values_dev, float *rewards_dev, uint8_t *dones_dev, uint64_t reset_seed, void *stream) {
    return mid_head_step_impl(v, h_dev, ld_h, in_features, packed_mid_dev, mid_features, packed_head_dev, seed, counter, actions_dev, action_dtype, logp_dev,
                              entropy_dev, values_dev, rewards_dev, dones_dev, &reset_seed, stream);
}

}  // extern "C"
