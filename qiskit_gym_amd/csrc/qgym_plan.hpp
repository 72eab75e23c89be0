// qgym_plan.hpp -- which layout a handle gets and which kernel a call launches, as pure host functions.
//
// The library has one kernel family per (env kind, size class, option set); the choices below are made from sizes and flags only, so they
// are written once, here, and used twice: by the code that allocates and launches (qgym_api.cpp, kernels_*.hip) and by qg_plan_query
// (include/qgym.h), which answers "what would run" without a GPU -- tests/test_dispatch.py pins every row of DESIGN.md's kernel table with
// it, so a changed threshold turns a CPU test red instead of silently routing a configuration to a slower-but-correct kernel.
#pragma once

#include <stdint.h>

#include "qgym_internal.hpp"

namespace qg {

enum Layout { LAYOUT_NONE = 0, LAYOUT_LF8 = 2, LAYOUT_PERM = 3, LAYOUT_PAULI = 4, LAYOUT_TILE = 5, LAYOUT_TILE64 = 6, LAYOUT_PERMB = 7, LAYOUT_LFD = 8 };

namespace plan {

// ---- limits -------------------------------------------------------------------------------------------------
constexpr uint32_t MAX_PERMUTATION_QUBITS = 256;   // one byte per entry
constexpr uint32_t MAX_LINEAR_FUNCTION_QUBITS = 64;  // one uint64 per row
constexpr uint32_t MAX_CLIFFORD_QUBITS = 32;       // 2N <= 64 columns
constexpr uint32_t MAX_PAULI_QUBITS = 32;
constexpr uint32_t MAX_PAULI_ROTATIONS = 32;

struct HandlePlan {
    Layout layout = LAYOUT_NONE;
    uint32_t D = 0;        // matrix rows (N for LinearFunction / Permutation)
    uint32_t nxp = 0;      // TILE: X-row slots; TILE64: row slots; PERMB / LFD: 16-byte groups (per region)
    bool has_z = false;    // Z-type rows present (CliffordEnv)
    bool w64 = false;      // LFD: uint64 rows
    uint32_t flags = 0;    // F_INVERTS | F_TRACK | F_LAYERS
    bool has_bad = false;  // the one-step kernels keep `solved` as an incremental per-env mask
    bool has_done_list = false;
    size_t stride_bytes = 0, state_bytes = 0;
    // PauliEnv
    uint32_t rmax = 0, rmax_generate = 0, pt_nq = 0, pt_rm = 0;
    int32_t max_rotations = 0;
    bool pauli_compact = false;
};

// The constructor's decisions (qg_vec_create).  Returns QG_OK, or the status and message the constructor reports.
inline int handle_plan(const qg_config &cfg, uint64_t batch, HandlePlan &p, const char *&why) {
    why = "";
    p = HandlePlan();
    const uint32_t N = (uint32_t)cfg.num_qubits;
    switch (cfg.env_kind) {
    case QG_PERMUTATION:
        if (N > MAX_PERMUTATION_QUBITS) { why = "PermutationEnv: N <= 256 supported (one byte per entry)"; return QG_ERR_UNSUPPORTED; }
        p.D = N;
        if (N <= 16) {  // one uint64 of nibbles per env
            p.layout = LAYOUT_PERM;
            p.stride_bytes = 8;
        } else {  // one byte per entry, tiles of 64 envs (kernels_perm.hip)
            p.layout = LAYOUT_PERMB;
            p.nxp = (N + 15u) / 16u;
            p.state_bytes = ((batch + 63) / 64) * (size_t)p.nxp * 1024;
        }
        break;
    case QG_LINEAR_FUNCTION:
        p.D = N;
        if (N > MAX_LINEAR_FUNCTION_QUBITS) { why = "LinearFunctionEnv: N <= 64 supported"; return QG_ERR_UNSUPPORTED; }
        if (N <= 8) {  // the whole matrix in one uint64
            p.layout = LAYOUT_LF8;
            p.stride_bytes = 8;
        }
        break;
    case QG_CLIFFORD:
        p.D = 2 * N;
        if (N > MAX_CLIFFORD_QUBITS) { why = "CliffordEnv: N <= 32 supported"; return QG_ERR_UNSUPPORTED; }
        break;
    case QG_PAULI: {
        p.D = 2 * N;
        p.layout = LAYOUT_PAULI;
        if (N > MAX_PAULI_QUBITS) { why = "PauliEnv: N <= 32 supported"; return QG_ERR_UNSUPPORTED; }
        const int max_rot = cfg.max_rotations > 1 ? cfg.max_rotations : 1;  // pauli.rs:387
        const int final_layers = cfg.final_pauli_layers >= 0 ? cfg.final_pauli_layers : cfg.max_rotations + 2;  // :760
        const int rmax = max_rot > final_layers ? max_rot : final_layers;
        if (rmax > (int)MAX_PAULI_ROTATIONS) { why = "PauliEnv: at most 32 rotations per env supported"; return QG_ERR_UNSUPPORTED; }
        p.rmax = (uint32_t)rmax;
        p.rmax_generate = (uint32_t)final_layers;  // reset() generates at most final_pauli_layers rotations (pauli.rs:563)
        p.max_rotations = max_rot;
        p.pt_nq = (N + 3u) & ~3u;
        p.pt_rm = p.rmax <= 8 ? 8u : p.rmax <= 16 ? 16u : 32u;
        p.pauli_compact = p.pt_nq <= 24 && p.pt_rm == 8;  // PTLayout::COMPACT: 12-byte qubit records, transposed rotation bits
        p.state_bytes = ((batch + 63) / 64) * ((size_t)p.pt_nq * (p.pauli_compact ? 768 : 1024) + (size_t)p.pt_rm * (p.pauli_compact ? 512 : 1024) +
                                               (p.pt_rm > 16 ? 3072 : 1024));  // PTLayout::TILE_BYTES
        break;
    }
    default: why = "unknown env_kind"; return QG_ERR_INVALID;
    }
    const bool inverts = cfg.add_inverts && cfg.env_kind != QG_PAULI;
    if (p.layout == LAYOUT_NONE && cfg.env_kind == QG_LINEAR_FUNCTION && inverts) {
        // the matrix and its inverse side by side: inversion is a role swap (kernels_lfd.hip)
        p.layout = LAYOUT_LFD;
        p.w64 = N > 32;
        const uint32_t rpg = p.w64 ? 2u : 4u;
        p.nxp = (N + rpg - 1u) / rpg;  // groups per matrix
        p.state_bytes = ((batch + 63) / 64) * (size_t)2 * p.nxp * 1024;
    }
    if (p.layout == LAYOUT_NONE && p.D <= 32) {  // thread-per-env TILE layout, uint32 rows (kernels_qm.hip): the hot path
        p.layout = LAYOUT_TILE;
        p.nxp = (N + 3u) & ~3u;
        p.has_z = cfg.env_kind == QG_CLIFFORD;
        const size_t R = p.has_z ? 2 * p.nxp : p.nxp;
        p.state_bytes = ((batch + 63) / 64) * R * 256;
    }
    if (p.layout == LAYOUT_NONE) {  // uint64 rows, thread per env (kernels_qm64.hip)
        p.layout = LAYOUT_TILE64;
        p.has_z = cfg.env_kind == QG_CLIFFORD;
        p.nxp = p.has_z ? 2u * ((N + 3u) & ~3u) : ((N + 7u) & ~7u);  // row slots per env
        p.state_bytes = ((batch + 63) / 64) * (size_t)p.nxp * 512;
    }
    if (inverts) p.flags |= F_INVERTS;
    if (cfg.track_solution) p.flags |= F_TRACK;
    if (!(cfg.w_n_layers == 0.0f && cfg.w_n_layers_cnots == 0.0f)) p.flags |= F_LAYERS;
    // TILE / TILE64 without add_inverts: `solved` as a per-env mask; PERMB: number of misplaced entries; LFD: row masks of both regions
    p.has_bad = ((p.layout == LAYOUT_TILE || p.layout == LAYOUT_TILE64) && !(p.flags & F_INVERTS)) || p.layout == LAYOUT_PERMB || p.layout == LAYOUT_LFD;
    p.has_done_list = p.layout == LAYOUT_TILE || p.layout == LAYOUT_TILE64 || p.layout == LAYOUT_PAULI;
    if (!p.state_bytes) p.state_bytes = p.stride_bytes * batch;
    return QG_OK;
}

// ---- step kernels ------------------------------------------------------------------------------------------------------------------
enum StepKernel {
    SK_INVALID = 0,
    SK_QM_STEP1,       // TILE one-step, no inverts: gathers / scatters the gate's <= 2 row groups (the headline kernel)
    SK_QM_INV2,        // TILE, add_inverts, every env symplectic: two lanes per env, inverse = bit transpose
    SK_QM_STEP_INV,    // TILE, add_inverts inside a fused rollout: thread per env, symplectic shortcut
    SK_QM_STEP_GJ,     // TILE, add_inverts, some env not known to be symplectic: Gauss-Jordan
    SK_QM_FUSED_LDS,   // TILE fused rollout, rows in LDS
    SK_QM_STEP,        // TILE register-resident (fused with features, empty gateset)
    SK_Q64_STEP1, SK_Q64_INV2, SK_Q64_STEP_INV, SK_Q64_STEP_GJ, SK_Q64_FUSED_LDS, SK_Q64_STEP,
    SK_LFD_STEP,       // LinearFunctionEnv with add_inverts: matrix + inverse
    SK_WORD_STEP,      // LF8 / PERM: one uint64 per env
    SK_PERMB_STEP1, SK_PERMB_STEP,
    SK_PTILE_STEP1C, SK_PTILE_STEP1, SK_PTILE_FUSED1C, SK_PTILE_STEP,
};
inline const char *step_kernel_name(StepKernel k) {
    switch (k) {
    case SK_QM_STEP1: return "qm_step1_kernel";
    case SK_QM_INV2: return "qm_inv2_kernel";
    case SK_QM_STEP_INV: return "qm_step_kernel<inv>";
    case SK_QM_STEP_GJ: return "qm_step_kernel<gauss-jordan>";
    case SK_QM_FUSED_LDS: return "qm_fused_lds_kernel";
    case SK_QM_STEP: return "qm_step_kernel";
    case SK_Q64_STEP1: return "q64_step1_kernel";
    case SK_Q64_INV2: return "q64_inv2_kernel";
    case SK_Q64_STEP_INV: return "q64_step_kernel<inv>";
    case SK_Q64_STEP_GJ: return "q64_step_kernel<gauss-jordan>";
    case SK_Q64_FUSED_LDS: return "q64_fused_lds_kernel";
    case SK_Q64_STEP: return "q64_step_kernel";
    case SK_LFD_STEP: return "lfd_step_kernel";
    case SK_WORD_STEP: return "word_step_kernel";
    case SK_PERMB_STEP1: return "permb_step1_kernel";
    case SK_PERMB_STEP: return "permb_step_kernel";
    case SK_PTILE_STEP1C: return "ptile_step1c_kernel";
    case SK_PTILE_STEP1: return "ptile_step1_kernel";
    case SK_PTILE_FUSED1C: return "ptile_fused1c_kernel";
    case SK_PTILE_STEP: return "ptile_step_kernel";
    default: return "(none)";
    }
}

// one launch of `T` steps (T == 1: env.step(); the graph rollouts issue T of those) on a TILE handle (kernels_qm.hip launch_step)
inline StepKernel tile_step(uint32_t flags, uint32_t T, bool has_bad, bool seq_outputs, uint32_t num_actions, bool has_z, uint32_t nxp) {
    const bool feat = flags & (F_TRACK | F_LAYERS);
    const bool seq = T != 1 || seq_outputs;
    if (has_bad && T == 1 && !(flags & F_INVERTS)) return SK_QM_STEP1;
    if (flags & F_INVERTS) {
        if (!(has_z && nxp <= 16)) return SK_INVALID;  // (LinearFunctionEnv with add_inverts lives in the LFD layout)
        if (!(flags & F_GJ) && T == 1) return SK_QM_INV2;
        return (flags & F_GJ) ? SK_QM_STEP_GJ : SK_QM_STEP_INV;
    }
    if (!feat && seq && T > 1 && num_actions != 0) return SK_QM_FUSED_LDS;
    return SK_QM_STEP;
}
// ... on a TILE64 handle (kernels_qm64.hip q64_launch_step)
inline StepKernel tile64_step(uint32_t flags, uint32_t T, bool has_bad, bool seq_outputs, uint32_t num_actions, bool has_z) {
    const bool feat = flags & (F_TRACK | F_LAYERS);
    const bool seq = T != 1 || seq_outputs;
    if (has_bad && T == 1 && !(flags & F_INVERTS)) return SK_Q64_STEP1;
    if (flags & F_INVERTS) {
        if (!has_z) return SK_INVALID;
        if (!(flags & F_GJ) && T == 1) return SK_Q64_INV2;
        return (flags & F_GJ) ? SK_Q64_STEP_GJ : SK_Q64_STEP_INV;
    }
    if (!feat && seq && T > 1 && num_actions != 0) return SK_Q64_FUSED_LDS;
    return SK_Q64_STEP;
}
inline StepKernel permb_step_kernel_of(uint32_t flags, uint32_t T, bool has_bad) {
    return (T == 1 && !(flags & F_INVERTS) && has_bad) ? SK_PERMB_STEP1 : SK_PERMB_STEP;
}
inline StepKernel pauli_step_kernel_of(uint32_t flags, uint32_t T, bool compact, bool has_perms) {
    const bool feat = flags & (F_TRACK | F_LAYERS);
    if (T == 1) return compact ? SK_PTILE_STEP1C : SK_PTILE_STEP1;
    if (!feat && !has_perms && compact) return SK_PTILE_FUSED1C;
    return SK_PTILE_STEP;
}
inline StepKernel step_kernel_of(const HandlePlan &p, uint32_t T, bool fused, bool seq_outputs, bool maybe_nonsymplectic, uint32_t num_actions, bool has_perms) {
    const uint32_t flags = p.flags | (maybe_nonsymplectic ? F_GJ : 0u);
    const uint32_t launch_T = fused ? T : 1u;  // a graph rollout is T single-step launches
    switch (p.layout) {
    case LAYOUT_TILE: return tile_step(flags, launch_T, p.has_bad, seq_outputs, num_actions, p.has_z, p.nxp);
    case LAYOUT_TILE64: return tile64_step(flags, launch_T, p.has_bad, seq_outputs, num_actions, p.has_z);
    case LAYOUT_LFD: return SK_LFD_STEP;  // (its fused form is T launches too)
    case LAYOUT_LF8:
    case LAYOUT_PERM: return SK_WORD_STEP;
    case LAYOUT_PERMB: return permb_step_kernel_of(flags, launch_T, p.has_bad);
    case LAYOUT_PAULI: return pauli_step_kernel_of(flags, launch_T, p.pauli_compact, has_perms);
    default: return SK_INVALID;
    }
}

// ---- qg_vec_reset_done on a list of finished envs (TILE / TILE64; decided on the DEVICE from the list's length) ------------------------
enum ResetPath { RP_FLAT = 0, RP_COOP = 1, RP_TREE = 2 };
inline const char *reset_path_name(ResetPath r) { return r == RP_TREE ? "scramble_tree" : r == RP_COOP ? "scramble_coop" : "scramble_flat"; }
constexpr uint32_t COOP_LANES = 16;        // lanes per env of the cooperative scramble (scramble_coop)
constexpr uint32_t TREE_THREADS = 256;     // one workgroup per env (scramble_tree)
constexpr uint32_t TREE_MAX_ENVS = 4096;   // four waves per env: the chip's SIMDs hold several of these waves each and issue slots, not the chain, set the time --
                                           // CliffordGym 16q, 262 144 envs: 4 096 finished 41 us as trees, 8 192 finished 48 us with 16 lanes each (tools/bench_reset_fraction.py)
constexpr uint32_t TREE_GRID = 1024;       // workgroups a launch sets aside for trees; a longer list is walked in rounds (a workgroup that finds no entry still runs the
                                           // kernel's prologue: 2 048 of them instead of 1 024 cost qg_vec_reset_done_step 2.3 us per pair at 65 536 envs, 4 096 cost
                                           // qg_vec_reset_done 3.5 us)
constexpr uint32_t TREE_MIN_DRAWS = 64;    // shorter chains do not repay the products
// lists of at most B / 32 finished envs take the 16-lanes-per-env path (count * 16 <= B / 2 threads)
__host__ __device__ inline bool coop_takes(uint32_t count, uint64_t B) { return (uint64_t)count * COOP_LANES * 2 <= B; }
// lists this short, of scrambles this long, go to scramble_tree: the one-lane-per-env form costs ~100 us whatever the list's length (its chain is
// `difficulty` gates long), a tree 15 us + ~4.5 us per 1 000 envs
__host__ __device__ inline bool tree_takes(uint32_t count, uint32_t n_draws) { return n_draws >= TREE_MIN_DRAWS && count <= TREE_MAX_ENVS; }
// workgroups of the launch that walk the list as trees (InitArgs::tree_grid); B / 8 keeps the grid in proportion to a small batch
inline uint32_t tree_grid(uint64_t B) { return (uint32_t)(B / 8u < TREE_GRID ? B / 8u : TREE_GRID); }
// PauliEnv (kernels_pauli_tile.hip): a workgroup per listed env (ptile_reset_tree_kernel) for lists up to B / 32 of scrambles this long.  No
// TREE_MAX_ENVS here: the per-lane generator is so much slower (130 us against 50 at 1 % of 65 536 envs) that the tree wins all the way.
// `n_cx`: CX gates in the gateset (the scramble draws from them; their table must fit the kernel's LDS copy).
constexpr uint32_t PAULI_CX_LDS = 1024;  // pairs (every ordered pair of 32 qubits is 992)
__host__ __device__ inline bool pauli_tree_takes(uint32_t count, uint32_t difficulty, uint64_t B, uint32_t n_cx) {
    return n_cx != 0 && n_cx <= PAULI_CX_LDS && difficulty >= TREE_MIN_DRAWS && coop_takes(count, B);
}
// `coop`: the host allows the cooperative paths (RNG draws, a row-operation table, B >= 64: InitArgs::coop); `coop_fits`: scramble_coop's LDS fits
__host__ __device__ inline ResetPath list_reset_path(uint32_t count, uint32_t n_draws, uint64_t B, bool coop, bool coop_fits) {
    if (coop && tree_takes(count, n_draws)) return RP_TREE;
    if (coop && coop_fits && coop_takes(count, B)) return RP_COOP;
    return RP_FLAT;
}
inline bool reset_coop_allowed(bool draws_given, uint64_t B, bool has_rowops) { return !draws_given && B >= 64 && has_rowops; }
// scramble_coop's LDS (4 waves x 4 envs x (R rows + 64 gate words)) against the init kernel's row array (4 waves x R x 64 words)
constexpr bool tile_coop_fits(uint32_t R, uint32_t word_bytes) { return 16ull * (R * word_bytes + 256ull) <= 4ull * R * 64ull * word_bytes; }

// qg_vec_reset_done_step as ONE launch (qm_reset_step_kernel): handles whose env.step() is the one-step TILE kernel
// (add_inverts keeps the two launches: a one-launch form -- the step on two lanes per env beside the reset workgroups, a reset env's first step on two
// lanes of the wave that wrote its fresh episode -- was built, parity-tested and measured at 15.9 - 18.9 us a pair against 14.3: 1 150 - 1 536 workgroups of 40 KB
// of LDS at three per CU, the step's 512 wait for slots behind the resets; as two graph branches a pair costs 10 us of fork and join.  EXPERIMENTS.md, round 5)
// ... and CliffordEnv N <= 16 with add_inverts (qm_reset_inv2_step_kernel: two lanes per env), while every env is known to be symplectic
inline bool reset_step_fusable(const HandlePlan &p) {
    if (!p.has_done_list) return false;
    if (p.flags & F_INVERTS) return p.layout == LAYOUT_TILE && p.has_z && p.nxp <= 16;
    return (p.layout == LAYOUT_TILE || p.layout == LAYOUT_TILE64) && p.has_bad;
}
// ... and whether the one launch is the faster form for a steady collection with this scramble length (`draws` = difficulty) and episode length (steps):
// what it gains is the tree resets' launch; when a large share of the batch finishes in every step (short episodes: the lane-per-env resets, whose envs' first steps
// read their episodes back) the two launches win -- with add_inverts as soon as the resets are not trees (31 against 19 us a pair at 50 % finishing), without when
// more than a quarter of the batch finishes per step (22 against 16); 64-bit rows also behind the 16-lane resets (18.1 against 16.6 at 3 % of Clifford 24q).
// From the CONFIGURATION, not from past list lengths: a captured launch keeps its choice for every replay.  (tools/probe_short_episodes.py)
inline bool reset_step_pays(const HandlePlan &p, int64_t draws, int64_t episode_steps) {
    if (p.flags & F_INVERTS) return draws >= (int64_t)TREE_MIN_DRAWS;
    if (p.layout == LAYOUT_TILE64 && draws < (int64_t)TREE_MIN_DRAWS && episode_steps >= 32) return false;
    return episode_steps >= 4;
}
inline bool reset_step_fuses(const HandlePlan &p) { return reset_step_fusable(p); }
// ... and on the one-word layouts (word_reset_step_kernel: the wave tests its envs' is_final flags itself, no list)
inline bool reset_step_in_word_kernel(const HandlePlan &p, size_t num_actions) { return (p.layout == LAYOUT_LF8 || p.layout == LAYOUT_PERM) && num_actions != 0; }

// ---- observations / state export (TILE) ----------------------------------------------------------------------------------------------
enum ExportKernel { EK_GENERIC = 0, EK_DENSE_STREAM, EK_DENSE_STREAM_ANY, EK_PACK, EK_WORDS_THEN_EXPAND };
inline const char *export_kernel_name(ExportKernel k) {
    return k == EK_DENSE_STREAM ? "qm_dense_stream_kernel" : k == EK_DENSE_STREAM_ANY ? "qm_dense_stream_any_kernel" : k == EK_PACK ? "qm_pack_kernel" : k == EK_WORDS_THEN_EXPAND ? "row words + expand" : "export_kernel";
}
// kernels_qm.hip qm_export: `R` = row slots of the layout (2 nxp with Z rows)
inline ExportKernel tile_export(uint32_t format, uint32_t D, uint32_t R, uint64_t out_stride, bool aligned16, bool aligned4) {
    if (format == QG_FMT_U8 && D == R && (D == 16 || D == 32) && out_stride == (uint64_t)D * D && aligned16) return EK_DENSE_STREAM;
    if (format == QG_FMT_U8 && D <= 32 && out_stride == (uint64_t)D * D && aligned16) return EK_DENSE_STREAM_ANY;  // any other size of the layout
    if (format == QG_FMT_PACKED && out_stride == D && D <= 32 && aligned4) return EK_PACK;
    return EK_GENERIC;
}
// qg_vec_get_state / set_state in the entry formats: through row words / a bit stream + a streaming kernel (qgym_api.cpp)
inline bool entry_formats_stream(Layout layout, uint64_t batch) {
    return batch >= QG_STREAM_MIN_ENVS && (layout == LAYOUT_TILE || layout == LAYOUT_TILE64 || layout == LAYOUT_LFD);
}
// qg_vec_track_dense: which handles can keep a resident dense observation, and whether the step kernel maintains it itself
inline bool dense_trackable(const HandlePlan &p) {
    const uint32_t R = p.has_z ? 2 * p.nxp : p.nxp;
    return p.layout == LAYOUT_TILE && p.D == R && (p.D == 16 || p.D == 32);
}
// ... which step kernels rewrite the rows they changed themselves: the one-step kernel without add_inverts, and CliffordEnv 16q's
// two-lanes-per-env kernel (whole env when the coin inverted it); every other launch is followed by a full rewrite
inline bool dense_in_kernel(const HandlePlan &p, StepKernel k) {
    return dense_trackable(p) && (k == SK_QM_STEP1 || (k == SK_QM_INV2 && p.has_z && p.D == 32));
}
inline bool dense_rides_in_step(const HandlePlan &p) {  // env.step() on a handle whose states are known to be symplectic
    return dense_in_kernel(p, step_kernel_of(p, 1, false, false, false, 1, false));
}

}  // namespace plan
}  // namespace qg
