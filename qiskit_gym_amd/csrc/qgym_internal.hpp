// qgym_internal.hpp -- shared host/device definitions of libqgym (gfx950 only).
//
// Resident state layouts (also documented in DESIGN.md section 3):
//
//  TILE layout (CliffordEnv N<=16, LinearFunctionEnv 8<N<=32 without add_inverts; the hot
//    path): thread-per-env, envs in tiles of 64, each tile = R/4 row groups of 1 KiB holding one
//    uint4 (4 row slots) per lane -- see kernels_qm.hip.
//  TILE64 layout: the same with uint64 rows (two per 16-byte group) for CliffordEnv 16<N<=32 and
//    LinearFunctionEnv 32<N<=64 without add_inverts -- see kernels_qm64.hip.
//  LFD layout  (LinearFunctionEnv 8<N<=64 with add_inverts): the matrix AND its inverse, inversion = a role swap -- see
//    kernels_lfd.hip.
//  LF8 layout   (LinearFunctionEnv N<=8): one uint64 per env, byte r = row r.
//  PERM layout  (PermutationEnv N<=16): one uint64 per env, nibble i = state[i].
//  PERMB layout (PermutationEnv 16<N<=256): one byte per entry, tiles of 64 envs x ceil(N/16) groups of 1 KiB -- see kernels_perm.hip.
//  PAULI layout (PauliEnv N<=32): lane q owns qubit q's tableau rows {X row q, Z row N+q} as two
//    uint64 (16 B), 32 lanes per env; rotations are 16-byte records {x mask, z mask, phase,
//    predecessor mask}, RMAX per env, owned by lanes 0..RMAX-1.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qgym.h"

namespace qg {

// ---- gate table entry: what one action does to the rows, plus its reward penalty ----------
// ops: two row operations, 14 bits each: [0:6) dst row, [6:12) src row, [12:14) type.
//   type 0 = none, 1 = dst ^= src, 2 = swap(dst, src).
// The host guarantees the two operations touch disjoint rows (clifford.rs:111-133: CX, CZ and
// SWAP with distinct qubits always do), so they commute.
// For PauliEnv the per-action program lives in its own table (kernels_pauli_tile.hip).
struct GateEntry {
    uint32_t ops;
    float penalty;  // metrics-weighted penalty of this action (metrics.rs:135-146), f32
};
static_assert(sizeof(GateEntry) == 8, "GateEntry must be 8 bytes");

enum : uint32_t { OP_NONE = 0, OP_XOR = 1, OP_SWAP = 2 };
__host__ __device__ inline uint32_t make_op(uint32_t type, uint32_t dst, uint32_t src) {
    return (dst & 63u) | ((src & 63u) << 6) | ((type & 3u) << 12);
}

// gate descriptor for the F_LAYERS metrics path and PauliEnv: kind | q0 << 8 | q1 << 16
__host__ __device__ inline uint32_t make_desc(uint32_t kind, uint32_t q0, uint32_t q1) {
    return (kind & 0xFFu) | ((q0 & 0xFFu) << 8) | ((q1 & 0xFFu) << 16);
}

// step-kernel behaviour flags
enum : uint32_t {
    F_ACT64 = 1u << 0,     // actions are int64
    F_INVERTS = 1u << 1,   // add_inverts (clifford.rs:262-270)
    F_TRACK = 1u << 2,     // track_solution (clifford.rs:334-340)
    F_LAYERS = 1u << 3,    // non-zero n_layers / n_layers_cnots weights: track per-qubit layers
    F_GJ = 1u << 4,        // some env may hold a non-symplectic matrix: compile in the Gauss-Jordan inversion
    F_DONE_LIST = 1u << 5  // the one-step kernel records the envs that finish for the qg_vec_reset_done that follows (no compaction launch): TILE stores
                           // one bit per env in StepArgs::done_mask, TILE64 and the sampling + step kernels append indices to StepArgs::done_list
};

// counter RNG shared by host, device and the tests
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t rng_draw(uint64_t seed, uint64_t env, uint64_t t) {
    return splitmix64(seed ^ splitmix64(env * 0x9E3779B97F4A7C15ull + t));
}
__host__ __device__ inline uint32_t rng_action(uint64_t seed, uint64_t env, uint64_t t, uint32_t num_actions) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__umul64hi(rng_draw(seed, env, t), (uint64_t)num_actions);
#else
    return (uint32_t)(((unsigned __int128)rng_draw(seed, env, t) * (unsigned __int128)num_actions) >> 64);
#endif
}

// Per-step / per-rollout kernel arguments common to every env kind.
struct StepArgs {
    void *state;
    const void *actions;      // [T][B]
    const uint8_t *coins;     // [T][B] or null
    const GateEntry *gates;   // [num_actions]
    const uint32_t *descs;    // [num_actions] kind | q0<<8 | q1<<16 (F_LAYERS, PauliEnv)
    int32_t *depth;
    float *reward;
    uint8_t *done;
    uint8_t *success;
    uint8_t *inverted;
    uint32_t *error;
    uint32_t *sol;            // [sol_cap][B] solution log, step-major (slots from the front: solution, from the back: solution_inv)
    int32_t *sol_len;         // [B][2]
    int32_t *layers;          // F_LAYERS: [B/64][2N + 2][64] last_gates, last_cxs, then (n_layers, n_layers_cnots) (layer_rec)
    float *rewards_seq;       // [T][B] or null
    uint8_t *dones_seq;       // [T][B] or null
    uint64_t B;
    uint64_t seed;            // counter-RNG seed for coins when coins == null
    uint64_t step_index;      // number of steps taken since creation (RNG counter)
    uint32_t D;               // rows (ROWS layout) / N (LF8, PERM)
    uint32_t N;               // qubits
    uint32_t log2L;
    uint32_t num_actions;
    uint32_t T;
    uint32_t flags;
    uint32_t sol_cap;
    float w[4];               // MetricsWeights, F_LAYERS only
    float pauli_layer_reward;
    uint32_t max_rotations;
    uint32_t kclk_waves;      // qg_vec_set_kernel_clock: wave records of this launch's slot
    const uint64_t *clock;    // device clock added to every RNG counter (qg_vec_set_clock), or null
    uint64_t env_base;        // global index of env 0 in the counter RNG (qg_vec_set_env_base)
    uint32_t *bad;            // TILE (uint32) / TILE64 (uint64) per-env mask: bit j = qubit j's rows / row j differ from the identity's; or null
    unsigned long long *kclk; // qg_vec_set_kernel_clock: this launch's slot -- kclk_waves records {entry, exit}, one per wave (device_common.hpp KernelClock) -- or null
    uint32_t *done_list;      // F_DONE_LIST: [B] indices of the envs that finished in this step, then {length, reader ticket} (compact_done's format);
    uint32_t *done_count;     // read only under that flag (the last fields of the block: other launches never touch their cache line)
    int8_t *dense;            // qg_vec_track_dense: the resident dense int8 observation [B][D][D]; the DENSE instantiations of the one-step kernels
                              // rewrite the rows their gate changed (read by those instantiations only)
    uint64_t *done_mask;      // F_DONE_LIST, TILE: word w = is_final of envs 64 w .. 64 w + 63 after this step, every word rewritten by the launch (no
                              // counter, nothing to zero: device_common.hpp done_mask_store); read by the next reset's workgroups (InitArgs::mask)
    uint32_t done_epoch;      // ... and a wave with a finisher stores this launch's number to the buffer's hint word: "not empty" (no device clock in it: the
                              // clock may advance between the step that writes and the reset that reads)
};

// The argument block spans four 64-byte lines and the scalar cache is cold at every launch.  Left alone, the compiler fetches a field
// right before its first use -- one cold line after the other along a one-step kernel's dependent chain (bounds check -> action -> gate
// entry -> rows), plus waited-for re-fetches in the middle of it.  Asking for every field the chain needs at the top of the kernel puts
// all misses in flight at once and keeps the values in SGPRs (CliffordEnv 16q x 65 536: 3.16 -> 3.10 us per step, same box).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QG_NO_ARG_PREFETCH)
#define QG_PREFETCH_STEP_ARGS(a)                                                                                                          \
    asm volatile("" ::"s"((a).state), "s"((a).actions), "s"((a).gates), "s"((a).depth), "s"((a).bad), "s"((a).B), "s"((a).flags),        \
                 "s"((a).num_actions), "s"((a).N), "s"((a).rewards_seq), "s"((a).dones_seq), "s"((a).reward), "s"((a).done), "s"((a).success), \
                 "s"((a).kclk))
#else
#define QG_PREFETCH_STEP_ARGS(a) ((void)0)
#endif

// PauliEnv's qg_vec_reset_done packs the finished envs' indices first (compact_done) only for batches above this: a small batch is a
// handful of waves that all hold a finished env anyway, and the extra launch costs more than the generator saves (the other envs
// always pack: their short lists go to the 16-lanes-per-env scramble, which is the faster one at any batch)
constexpr uint64_t QG_COMPACT_MIN_ENVS = 4096;
// set_state / get_state in the entry formats (Vec<i64>, dense bytes) go through a bit stream / row words plus a streaming kernel from this
// batch size on; below it (the scalar qg_env_* handles are batches of one) the single init / export launch is the cheaper one
constexpr uint64_t QG_STREAM_MIN_ENVS = 64;

// state (re)initialisation
struct InitArgs {
    void *state;
    int32_t *depth;
    float *reward;
    uint8_t *done;
    uint8_t *success;
    uint8_t *inverted;
    uint32_t *error;
    int32_t *sol_len;
    int32_t *layers;
    const GateEntry *gates;
    const int32_t *actions;   // reset_with draws [n_draws][B] or null (use RNG)
    const void *src;          // set_state source
    uint64_t src_stride;      // elements per env in src
    uint64_t B;
    uint64_t seed;
    uint32_t D, N, log2L, num_actions;
    uint32_t n_draws;
    int32_t depth_value;
    uint32_t format;          // qg_state_format
    uint32_t mode;            // 0 identity, 1 set_state, 2 scramble
    uint32_t layers_len;
    uint32_t check_symplectic; // TILE layout with add_inverts: record whether the state is symplectic
    uint32_t only_done;        // reset only the envs whose `done` flag is set (auto-reset between episodes)
    uint32_t flags_current;    // a launch of its own (qg_vec_reset_done, not inside a reset + step launch): nobody writes `done` meanwhile, so a LONG list's lane-per-env
                               // resets test the env's own flag (thread = env) instead of searching the mask / list for "entry i" (TILE)
    uint32_t *nonsymp_flag;    // set_state with add_inverts: or-ed to 1 when some env is not symplectic
    const uint64_t *clock;     // device clock: the scramble seed becomes seed + 0x9E3779B9 * clock (qg_vec_set_clock)
    uint32_t *bad;             // see StepArgs::bad
    const uint64_t *mask;      // reset_done, TILE: the finished envs as the bits a step launch left (StepArgs::done_mask), mask_words words; the envs to
    uint32_t mask_words;       // reset are the set bits in ascending order FOLLOWED by the list's entries (null: the list alone)
    uint32_t mask_epoch;       // StepArgs::done_epoch of the launch that wrote `mask`: a hint word that differs says "no bit is set" without a count
    uint32_t *count_pub;       // (unused by the init kernels: PauliEnv's two launches pass the mask's count through qg_vec::mask_count)
    const uint32_t *list;      // reset_done, compacted: thread i resets env list[i], i < *list_count (or null: thread = env)
    uint32_t *list_count;      // [2]: length, reader ticket (device_common.hpp list_count_take)
    uint32_t tree_grid;        // workgroups of this launch that walk a list as trees (plan::tree_grid), entry i on workgroup i mod tree_grid
    uint32_t *zero_count;      // [2] of ANOTHER list, idle during this launch, or null: zeroed here instead of a reader ticket on list_count (list_count_take)
    uint32_t coop;             // list mode with RNG draws: the 16-lanes-per-env scramble kernel handles small lists
    const uint32_t *rowops;    // TILE layout: per action two row operations (make_op, slot indices) for that kernel
    uint64_t env_base;         // global index of env 0 in the counter RNG (qg_vec_set_env_base)
    uint32_t inverts;          // add_inverts is set (PermutationEnv set_state: a state with a repeated entry is a fault only then)
    int8_t *dense;             // qg_vec_track_dense + a list of finished envs: the reset rewrites those envs' dense observation (else null: the host
                               // refreshes the whole buffer after the launch)
    unsigned long long *kclk;  // qg_vec_set_kernel_clock: this launch's slot (kclk_waves wave records), or null
    uint32_t kclk_waves;
    uint32_t *count_out;       // reset_done with a list / mask, TILE: the number of envs reset, for the host's eyes (pinned memory; sizes the next launches' tree grid), or null
};

// A handle may be given a device-resident clock (qg_vec_set_clock).  Launches replayed from a
// captured graph carry their RNG counters as baked-in kernel arguments; adding the clock, which the
// graph's owner advances between replays, keeps every replay's draws distinct and reproducible.
#define QG_CLOCK_SEED_STRIDE 0x9E3779B9ull
__device__ inline uint64_t clock_of(const uint64_t *c) { return c ? *c : 0ull; }
__device__ inline uint64_t step_clock(const StepArgs &a) { return a.step_index + clock_of(a.clock); }
__device__ inline uint64_t init_seed(const InitArgs &a) { return a.seed + QG_CLOCK_SEED_STRIDE * clock_of(a.clock); }

struct ObsArgs {
    const void *state;
    void *out;
    uint64_t B;
    uint64_t out_stride;      // elements per env in out
    uint32_t D, N, log2L;
    uint32_t format;          // qg_state_format (U8 -> int8 dense, I64, PACKED)
    uint32_t obs_rows, obs_cols;
    unsigned long long *kclk; // qg_vec_set_kernel_clock: this launch's slot (kclk_waves wave records), or null
    uint32_t kclk_waves;
};

// internal set_state source format (never crosses the C ABI): InitArgs::src is a bit stream, bit e = entry e of the flat [B][D][D] array
// (pack_bitstream, kernels_collect.hip); src_stride = D * D
#define QG_FMT_BITS 3u
// D <= 64 bits of the stream starting at bit `pos` (the word after the stream's last one must be readable: the host pads it)
__device__ inline uint64_t bits_window(const uint64_t *bs, uint64_t pos, uint32_t D) {
    const uint32_t sh = (uint32_t)pos & 63u;
    const uint64_t lo = bs[pos >> 6], hi = bs[(pos >> 6) + 1];
    const uint64_t w = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
    return D >= 64u ? w : w & ((1ull << D) - 1ull);
}

// launchers (one translation unit per layout)
// LinearFunctionEnv with add_inverts, 8 < N <= 64: state and inverse side by side, `rg` 16-byte groups per matrix (kernels_lfd.hip)
hipError_t lfd_step(const StepArgs &a, bool w64, uint32_t rg, hipStream_t s);
hipError_t lfd_init(const InitArgs &a, bool w64, uint32_t rg, const uint32_t *descs, hipStream_t s);
hipError_t lfd_export(const ObsArgs &a, bool w64, uint32_t rg, const uint8_t *inverted, hipStream_t s);

hipError_t qm_step(const StepArgs &a, uint32_t nxp, bool has_z, hipStream_t s);
hipError_t qm_init(const InitArgs &a, uint32_t nxp, bool has_z, hipStream_t s);
// qg_vec_reset_done (mask + list in `reset`) + qg_vec_step (`step`, F_DONE_LIST: writes ITS mask, a reset env that is final again after its first step
// goes to ITS list) in one launch
hipError_t qm_reset_step(const InitArgs &reset, const StepArgs &step, uint32_t nxp, bool has_z, hipStream_t s);
hipError_t q64_reset_step(const InitArgs &reset, const StepArgs &step, uint32_t ns, bool has_z, hipStream_t s);  // kernels_qm64.hip: the same for 64-bit rows
hipError_t qm_export(const ObsArgs &a, uint32_t nxp, bool has_z, hipStream_t s);
// dense {0,1} observation in an element type of `elem_size` bytes whose 1 is the bit pattern `one`
hipError_t qm_export_typed(const void *state, uint64_t B, uint32_t N, uint32_t D, uint32_t nxp, bool has_z, void *out, uint32_t elem_size,
                           uint32_t one, hipStream_t s);

hipError_t q64_step(const StepArgs &a, uint32_t ns, bool has_z, hipStream_t s);
hipError_t q64_init(const InitArgs &a, uint32_t ns, bool has_z, hipStream_t s);
hipError_t q64_export(const ObsArgs &a, uint32_t ns, bool has_z, hipStream_t s);

hipError_t lf8_step(const StepArgs &a, bool fused, hipStream_t s);
hipError_t lf8_init(const InitArgs &a, hipStream_t s);
hipError_t lf8_export(const ObsArgs &a, hipStream_t s);

// LF8 / PERM: qg_vec_reset_done (counter-RNG draws) + qg_vec_step in one launch (kernels_small.hip word_reset_step_kernel)
hipError_t word_reset_step(const InitArgs &reset, const StepArgs &step, bool perm, hipStream_t s);
hipError_t perm_step(const StepArgs &a, bool fused, hipStream_t s);
hipError_t perm_init(const InitArgs &a, hipStream_t s);
hipError_t perm_export(const ObsArgs &a, hipStream_t s);
// PermutationEnv with more than 16 qubits: one byte per entry, `ng` 16-byte groups per env (kernels_perm.hip)
hipError_t permb_step(const StepArgs &a, uint32_t ng, hipStream_t s);
hipError_t permb_init(const InitArgs &a, uint32_t ng, const uint32_t *descs, hipStream_t s);
hipError_t permb_export(const ObsArgs &a, uint32_t ng, hipStream_t s);

// dense {0,1} tensor of `out_dtype` (qg_dtype) from rows packed one per word (kernels_collect.hip)
hipError_t expand_rows(const void *words_dev, int word_bytes, uint64_t n_rows, uint32_t cols, void *out_dev, int out_dtype, hipStream_t s);
// the trait's Vec<i64> wire format: [n_rows * cols] int64 {0, 1} entries from row words (get_state), and the flat entry stream (int64 / int8,
// > 0 means 1) as a bit stream of 64-bit words for the init kernels' QG_FMT_BITS (set_state)
hipError_t expand_rows_i64(const void *words_dev, int word_bytes, uint64_t n_rows, uint32_t cols, int64_t *out_dev, hipStream_t s);
hipError_t pack_bitstream(const void *src, int elem_bytes, uint64_t n_entries, uint64_t *out_words, hipStream_t s);
hipError_t compact_done(const uint8_t *done, uint64_t B, uint32_t *list, uint32_t *count, hipStream_t s);
hipError_t masks_fill(const uint8_t *success, uint8_t *out, uint64_t B, uint32_t num_actions, hipStream_t s);
hipError_t fault_any(const uint32_t *error, uint64_t B, uint32_t *scratch, uint32_t *out_host, hipStream_t s);
hipError_t step_outputs(const float *reward, const uint8_t *done, const uint8_t *success, float *rewards_out, uint8_t *dones_out, uint8_t *success_out,
                        uint64_t B, hipStream_t s);

}  // namespace qg
