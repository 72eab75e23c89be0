// qgym_api.cpp -- host side of libqgym: the C ABI of include/qgym.h over the gfx950 kernels.
//
// What lives here is what the reference does in its constructors and trait plumbing:
//   gate parsing/validation        rust/src/envs/common.rs:46-100
//   constructor defaults           rust/src/envs/clifford.rs:401-426, pauli.rs:743-778
//   MetricsWeights / per-gate cost rust/src/envs/metrics.rs:64-81,135-185
//   Env trait method set           rust/src/envs/clifford.rs:285-382
// plus device memory ownership, stream-ordered launches and the hipGraph cache for rollouts.
// There is deliberately no CPU implementation of the env in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "pauli_common.hpp"

namespace qg {

static thread_local std::string g_err;

int set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

static uint32_t pow2ceil_log2(uint32_t x) {
    uint32_t l = 0;
    while ((1u << l) < x) ++l;
    return l;
}

// ---- per-gate metric deltas (metrics.rs:64-123 with its guards) -----------------------------
static void gate_deltas(const qg_gate &g, uint32_t N, int &dc, int &dg) {
    dc = dg = 0;
    const uint32_t a = (uint32_t)g.q0, b = (uint32_t)g.q1;
    auto single = [&](uint32_t t) { if (t < N) dg += 1; };
    auto cx = [&](uint32_t c, uint32_t t) { if (c != t && c < N && t < N) { dc += 1; dg += 1; } };
    switch (g.kind) {
    case QG_CX: cx(a, b); break;
    case QG_SWAP: cx(a, b); cx(b, a); cx(a, b); break;
    case QG_CZ: single(b); cx(a, b); single(b); break;
    default: single(a); break;
    }
}

// weighted_delta (metrics.rs:135-146) for an action whose layer deltas are multiplied by zero
// weights: f32, left to right.  Built with -ffp-contract=off.
static float table_penalty(const float w[4], int dc, int dg) {
    volatile float t0 = w[0] * (float)dc;
    volatile float t1 = w[1] * 1.0f;  // weight is +-0 here; keeps its sign like w * delta would
    volatile float t2 = w[2] * 1.0f;
    volatile float t3 = w[3] * (float)dg;
    volatile float s = t0 + t1;
    s = s + t2;
    s = s + t3;
    return s;
}

// row operations of one gate (clifford.rs:89-133; linear_function.rs:62-83,237-243; permutation.rs:205-208)
static uint32_t gate_ops(int env_kind, const qg_gate &g, uint32_t N) {
    const uint32_t a = (uint32_t)g.q0, b = (uint32_t)g.q1;
    switch (env_kind) {
    case QG_CLIFFORD:
        switch (g.kind) {
        case QG_H: return make_op(OP_SWAP, a, N + a);
        case QG_S:
        case QG_SDG: return make_op(OP_XOR, N + a, a);
        case QG_SX:
        case QG_SXDG: return make_op(OP_XOR, a, N + a);
        case QG_CX: return a == b ? 0u : make_op(OP_XOR, b, a) | (make_op(OP_XOR, N + a, N + b) << 14);
        case QG_CZ: return a == b ? 0u : make_op(OP_XOR, N + a, b) | (make_op(OP_XOR, N + b, a) << 14);
        case QG_SWAP: return a == b ? 0u : make_op(OP_SWAP, a, b) | (make_op(OP_SWAP, N + a, N + b) << 14);
        }
        return 0u;
    case QG_LINEAR_FUNCTION:
        if (g.kind == QG_CX) return a == b ? 0u : make_op(OP_XOR, b, a);
        if (g.kind == QG_SWAP) return a == b ? 0u : make_op(OP_SWAP, a, b);
        return 0u;
    case QG_PERMUTATION:
        if (g.kind == QG_SWAP) return a == b ? 0u : make_op(OP_SWAP, a, b);
        return 0u;
    }
    return 0u;
}

int ensure_scratch_public(qg_vec *v, size_t bytes) {
    if (bytes <= v->scratch_bytes) return QG_OK;
    if (v->scratch) {
        if (hipDeviceSynchronize() != hipSuccess || hipFree(v->scratch) != hipSuccess) {
            (void)hipGetLastError();
            return set_error(QG_ERR_DEVICE, "scratch release failed");
        }
        v->scratch = nullptr;
        v->scratch_bytes = 0;
    }
    hipError_t e = hipMalloc(&v->scratch, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    v->scratch_bytes = bytes;
    return QG_OK;
}

// TILE layout action word (kernels_qm.hip): q0, q1 and the 4x4 GF(2) matrix acting on
// {X[q0], Z[q0], X[q1], Z[q1]} (clifford.rs:89-133 / linear_function.rs:62-83 as linear maps)
static uint32_t tile_ops(int env_kind, const qg_gate &g, bool wide) {
    const uint32_t a = (uint32_t)g.q0, b = (uint32_t)g.q1;
    enum { X0 = 1, Z0 = 2, X1 = 4, Z1 = 8 };
    auto M = [](uint32_t ox0, uint32_t oz0, uint32_t ox1, uint32_t oz1) { return ox0 | (oz0 << 4) | (ox1 << 8) | (oz1 << 12); };
    const uint32_t ident = M(X0, Z0, X1, Z1);
    uint32_t m = ident, q1 = a;
    if (env_kind == QG_CLIFFORD) {
        switch (g.kind) {
        case QG_H: m = M(Z0, X0, X1, Z1); break;
        case QG_S:
        case QG_SDG: m = M(X0, Z0 | X0, X1, Z1); break;
        case QG_SX:
        case QG_SXDG: m = M(X0 | Z0, Z0, X1, Z1); break;
        case QG_CX: if (a != b) { m = M(X0, Z0 | Z1, X1 | X0, Z1); q1 = b; } break;
        case QG_CZ: if (a != b) { m = M(X0, Z0 | X1, X1, Z1 | X0); q1 = b; } break;
        case QG_SWAP: if (a != b) { m = M(X1, Z1, X0, Z0); q1 = b; } break;
        }
    } else {  // LinearFunction: rows are the X-type slots only
        if (g.kind == QG_CX && a != b) { m = M(X0, Z0, X1 | X0, Z1); q1 = b; }
        if (g.kind == QG_SWAP && a != b) { m = M(X1, Z0, X0, Z1); q1 = b; }
    }
    if (wide) return (a & 63u) | ((q1 & 63u) << 6) | (m << 12);  // TILE64 (kernels_qm64.hip)
    return (a & 31u) | ((q1 & 31u) << 5) | (m << 10);
}

}  // namespace qg

using namespace qg;

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *qg_last_error(void) { return g_err.c_str(); }
int qg_abi_version(void) { return QG_ABI_VERSION; }

int qg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void qg_config_default(qg_config *c, int32_t env_kind, int32_t num_qubits) {
    memset(c, 0, sizeof *c);
    c->env_kind = env_kind;
    c->num_qubits = num_qubits;
    c->difficulty = 1;   // envs/synthesis.py:186
    c->depth_slope = 2;  // :187
    c->max_depth = 128;  // :188
    c->w_n_cnots = 0.01f;       // metrics.rs:160
    c->w_n_layers_cnots = 0.0f; // :161
    c->w_n_layers = 0.0f;       // :162
    c->w_n_gates = 0.0001f;     // :163
    c->add_inverts = env_kind == QG_PAULI ? 0 : 1;  // clifford.rs:420 (PauliEnv has no inversion)
    c->add_perms = 1;                               // clifford.rs:421
    c->track_solution = 1;                          // clifford.rs:422
    c->max_rotations = 5;                           // envs/synthesis.py:387
    c->pauli_diff_scale = 16;                       // envs/synthesis.py:388 (Rust-side default 8)
    c->final_pauli_layers = -1;                     // None -> max_rotations + 2 (pauli.rs:760)
    c->num_qubits_decay = 0.5f;                     // pauli.rs:769
    c->pauli_layer_reward = 0.01f;                  // pauli.rs:773
}

int qg_gate_parse(const char *name_in, const int64_t *idx, size_t n, qg_gate *out) {
    if (!name_in || !out) return set_error(QG_ERR_TYPE, "Gate name must be a string");
    std::string name(name_in);
    size_t b = 0, e = name.size();
    while (b < e && isspace((unsigned char)name[b])) ++b;
    while (e > b && isspace((unsigned char)name[e - 1])) --e;
    name = name.substr(b, e - b);  // common.rs:64 trim
    std::string key = name;
    for (auto &ch : key) ch = (char)tolower((unsigned char)ch);  // :65
    for (size_t i = 0; i < n; ++i)
        if (idx[i] < 0) return set_error(QG_ERR_TYPE, "Gate indices must be non-negative integers (usize)");
    static const char *one_q[] = {"h", "s", "sdg", "sx", "sxdg"};
    static const char *two_q[] = {"cx", "cz", "swap"};
    for (int k = 0; k < 5; ++k)
        if (key == one_q[k]) {
            if (n != 1) return set_error(QG_ERR_INVALID, "Gate `%s` expects 1 index, got %zu", name.c_str(), n);
            out->kind = k;
            out->q0 = (int32_t)idx[0];
            out->q1 = 0;
            return QG_OK;
        }
    for (int k = 0; k < 3; ++k)
        if (key == two_q[k]) {
            if (n != 2) return set_error(QG_ERR_INVALID, "Gate `%s` expects 2 indices, got %zu", name.c_str(), n);
            out->kind = 5 + k;
            out->q0 = (int32_t)idx[0];
            out->q1 = (int32_t)idx[1];
            return QG_OK;
        }
    return set_error(QG_ERR_INVALID, "Unknown gate name `%s`. Allowed: H, S, Sdg, SX, SXdg, CX, CZ, SWAP", name.c_str());
}

// ------------------------------------------------------------------------------------------------
// qg_vec
// ------------------------------------------------------------------------------------------------

static int vec_free_buffers(qg_vec *v) {
    void *ptrs[] = {v->state, v->own_depth ? v->depth : nullptr, v->own_reward ? v->reward : nullptr,
                    v->own_done ? v->done : nullptr, v->own_success ? v->success : nullptr, v->inverted, v->own_error ? v->error : nullptr, v->sol,
                    v->sol_len, v->layers, v->d_gates, v->d_descs, v->scratch, v->d_prog,
                    v->d_qubit_perms, v->d_act_perms, v->perm_idx, v->d_gen_tables, v->d_nonsymp, v->bad, v->done_list, v->done_list_alt, v->done_list_spare, v->done_mask[0], v->done_mask[1], v->mask_count, v->d_rowops, v->embed_dump, v->host_in, v->host_obs, v->fault_scratch};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &g : v->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    v->graphs.clear();
    if (v->capture_stream) (void)hipStreamDestroy(v->capture_stream);
    if (v->fault_word) (void)hipHostFree(v->fault_word);
    if (v->count_seen) (void)hipHostFree(v->count_seen);
    return 0;
}

static int ensure_scratch(qg_vec *v, size_t bytes) { return qg::ensure_scratch_public(v, bytes); }

}  // extern "C"
namespace qg {
// ---- the list of finished envs (done_list: [B] indices, then {length, reader ticket}) --------------------------------------------
// Device-side facts: a LIST step kernel (and the sampling + step kernels) APPENDS to the list, so the length must be zero when it
// starts; the reset kernel that consumes a list zeroes the length again (list_count_take); compact_done zeroes it itself.  What the
// host knows about the length is exact only for launches it has enqueued eagerly, in order: anything captured into a caller's graph
// runs later, any number of times, between whatever else the caller enqueues.  So the host's belief (done_list_fresh, list_zero_known)
// is scoped to a SESSION -- one stream capture (its capture id), or eager execution on a handle none of whose list launches were
// ever captured:
//   * a new session starts with nothing known: its first appending launch is preceded by a memset of the length (captured with it),
//     its first qg_vec_reset_done compacts the `done` flags itself;
//   * once anything was captured (list_tainted), eager calls trust nothing: no LIST instantiations, every reset_done compacts;
//   * inside a session the launches run in the order they were enqueued, so the belief is exact there.
// Appends are clamped to the list's B entries on the device as well (done_list_append), so a misuse cannot write past the allocation.
bool done_list_session(qg_vec *v, hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (s) (void)hipStreamGetCaptureInfo(s, &cs, &id);
    const uint64_t cur = cs == hipStreamCaptureStatusNone ? 0ull : (id ? (uint64_t)id : ~0ull);
    if (cur != v->list_session) {
        v->list_session = cur;
        v->done_list_fresh = false;
        v->list_zero_known = false;
        v->mask_fresh = false;
        v->alt_zero_known = false;
        if (cur) v->list_tainted = true;
    }
    return cur != 0 || !v->list_tainted;
}

// before a launch that appends: the length is zero when it runs
int done_list_before_append(qg_vec *v, hipStream_t s) {
    if (!v->list_zero_known || v->done_list_fresh) HIP_TRY(hipMemsetAsync(v->done_list + v->B, 0, 2 * sizeof(uint32_t), s));
    v->done_list_fresh = false;
    v->mask_fresh = false;
    v->list_zero_known = true;
    return QG_OK;
}
void done_list_appended(qg_vec *v, bool trusted) {
    v->done_list_fresh = trusted;  // an untrusted list is never consumed: the next reset_done compacts
    v->list_zero_known = false;
    v->mask_fresh = false;  // (the step launches that left their finishers as a mask say so themselves: step_wrote_mask)
}
// the number a mask-writing launch stamps its buffer's hint word with (never 0, the buffers' initial content)
static uint32_t mask_epoch_of(const qg_vec *v) { return ((uint32_t)v->step_index & 0x7FFFFFFFu) + 1u; }
// TILE: a list-leaving step writes the mask that is not the current one (and appends nothing: the list stays empty)
static void step_wrote_mask(qg_vec *v, const StepArgs &a, bool trusted) {
    if (a.done_mask) {
        v->mask_cur ^= 1;
        v->mask_epoch[v->mask_cur] = a.done_epoch;
        v->mask_fresh = trusted;
        v->list_zero_known = true;  // (nothing was appended: the length is still the zero done_list_before_append made sure of)
    }
}

}  // namespace qg
extern "C" {
// A list describes the `done` flags of the step that wrote it only: anything else that changes the flags first drops it (and re-zeroes
// its length, which only the list's consumer would have done).
static int drop_done_list(qg_vec *v, hipStream_t s) {
    (void)done_list_session(v, s);
    if (v->done_list_fresh) {
        if (!(v->mask_fresh && v->list_zero_known)) HIP_TRY(hipMemsetAsync(v->done_list + v->B, 0, 2 * sizeof(uint32_t), s));
        v->done_list_fresh = false;
        v->list_zero_known = true;
    }
    v->mask_fresh = false;
    return QG_OK;
}

// qg_vec_set_kernel_clock: the slot of the launch about to be enqueued (the k-th one after the call), or null
static unsigned long long *kernel_clock_slot(const qg_vec *v) {
    if (!v->kclk || v->kclk_next >= v->kclk_cap) return nullptr;
    return v->kclk + 2ull * v->kclk_waves * (v->kclk_next++);
}

// Workgroups a reset_done launch sets aside for trees (InitArgs::tree_grid).  Every one of them costs a dispatch slot and ~2 us of a CU's third of its
// LDS whether or not the list reaches it, and at three workgroups per CU whatever is dispatched late waits -- so the grid follows the list lengths
// this handle's resets have reported (InitArgs::count_out: the latest launch that has finished; read without waiting): the length, four standard
// deviations of a count that size and a margin, in steps of 64.  A longer list is walked in rounds: the grid's size never changes a result.
static uint32_t reset_tree_grid(const qg_vec *v, uint32_t most) {
    const uint32_t seen = v->count_seen ? *(volatile const uint32_t *)v->count_seen : 0xFFFFFFFFu;
    if (seen == 0xFFFFFFFFu) return most;
    // A launch being CAPTURED keeps its grid for every replay (eager calls correct themselves at the next call): a capture made while few envs finish -- right
    // after a reset, say -- must not bake a small grid in, or every later list is walked in many rounds.  TILE: the whole grid (its one-launch kernels fit
    // seven workgroups per CU since round 5: idle tree workgroups cost nothing measurable, 9.15 against 9.2 us a pair); TILE64 / PauliEnv (three per CU: the
    // whole grid costs 1.0 / 0.5 us a pair): at least half of it, all of it when the last launch saw nobody finish.
    const bool captured = v->list_session != 0;
    if (captured && (v->layout == LAYOUT_TILE || seen == 0)) return most;
    const uint64_t want = (uint64_t)seen + 4ull * (uint64_t)std::sqrt((double)seen) + 32ull;
    const uint64_t g = (want + 63ull) & ~63ull;
    return (uint32_t)std::min<uint64_t>(most, std::max<uint64_t>(captured ? std::max<uint64_t>(64ull, most / 2) : 64ull, g));
}


static void fill_init_args(const qg_vec *v, InitArgs &a) {
    memset(&a, 0, sizeof a);
    a.tree_grid = plan::tree_grid(v->B);
    a.state = v->state;
    a.depth = v->depth;
    a.reward = v->reward;
    a.done = v->done;
    a.success = v->success;
    a.inverted = v->inverted;
    a.error = v->error;
    a.sol_len = v->sol_len;
    a.layers = v->layers;
    a.layers_len = v->layers_len;
    a.gates = v->d_gates;
    a.B = v->B;
    a.D = v->D;
    a.N = v->N;
    a.log2L = v->log2L;
    a.num_actions = (uint32_t)v->gates.size();
    a.clock = v->clock_dev;
    a.env_base = v->env_base;
    a.bad = v->bad;
    a.rowops = v->d_rowops;
    a.inverts = (v->flags & F_INVERTS) ? 1u : 0u;
    a.check_symplectic = ((v->layout == LAYOUT_TILE || v->layout == LAYOUT_TILE64) && (v->flags & F_INVERTS)) ? 1u : 0u;
}

static hipError_t launch_init(const qg_vec *v, const InitArgs &a_in, hipStream_t s) {
    InitArgs a = a_in;
    a.kclk = kernel_clock_slot(v);
    a.kclk_waves = v->kclk_waves;
    switch (v->layout) {
    case LAYOUT_LFD: return lfd_init(a, v->w64, v->nxp, v->d_descs, s);
    case LAYOUT_LF8: return lf8_init(a, s);
    case LAYOUT_PERM: return perm_init(a, s);
    case LAYOUT_PERMB: return permb_init(a, v->nxp, v->d_descs, s);
    case LAYOUT_TILE: return qm_init(a, v->nxp, v->has_z, s);
    case LAYOUT_TILE64: return q64_init(a, v->nxp, v->has_z, s);
    default: return hipErrorInvalidValue;
    }
}

static void fill_step_args(const qg_vec *v, StepArgs &a) {
    memset(&a, 0, sizeof a);
    a.state = v->state;
    a.gates = v->d_gates;
    a.descs = v->d_descs;
    a.depth = v->depth;
    a.reward = v->reward;
    a.done = v->done;
    a.success = v->success;
    a.inverted = v->inverted;
    a.error = v->error;
    a.sol = v->sol;
    a.sol_len = v->sol_len;
    a.layers = v->layers;
    a.B = v->B;
    a.seed = v->coin_seed;
    a.step_index = v->step_index;
    a.clock = v->clock_dev;
    a.env_base = v->env_base;
    a.bad = v->bad;
    a.D = v->D;
    a.N = v->N;
    a.log2L = v->log2L;
    a.num_actions = (uint32_t)v->gates.size();
    a.T = 1;
    a.flags = v->flags | (v->maybe_nonsymplectic ? F_GJ : 0u);
    a.sol_cap = v->sol_cap;
    a.w[0] = v->cfg.w_n_cnots;
    a.w[1] = v->cfg.w_n_layers_cnots;
    a.w[2] = v->cfg.w_n_layers;
    a.w[3] = v->cfg.w_n_gates;
    a.pauli_layer_reward = v->cfg.pauli_layer_reward;
    a.max_rotations = (uint32_t)v->cfg.max_rotations;
}

static hipError_t launch_step(const qg_vec *v, const StepArgs &a_in, hipStream_t s) {
    StepArgs a = a_in;
    a.kclk = kernel_clock_slot(v);
    a.kclk_waves = v->kclk_waves;
    switch (v->layout) {
    case LAYOUT_LFD: return lfd_step(a, v->w64, v->nxp, s);
    case LAYOUT_LF8: return lf8_step(a, a.T > 1, s);
    case LAYOUT_PERM: return perm_step(a, a.T > 1, s);
    case LAYOUT_PERMB: return permb_step(a, v->nxp, s);
    case LAYOUT_PAULI: return pauli_step(v, a, s);
    case LAYOUT_TILE: return qm_step(a, v->nxp, v->has_z, s);
    case LAYOUT_TILE64: return q64_step(a, v->nxp, v->has_z, s);
    default: return hipErrorInvalidValue;
    }
}

// qg_vec_track_dense.  The one-step kernel without add_inverts (qm_step1_kernel) rewrites the rows its gate changed in the same launch;
// every other launch that changes states is followed by a full rewrite (dense_refresh), except resets of a list of finished envs, which
// rewrite those envs' observations themselves.
static plan::HandlePlan plan_of(const qg_vec *v) {  // the handle's fields that the planner's predicates read
    plan::HandlePlan hp;
    hp.layout = v->layout;
    hp.D = v->D;
    hp.nxp = v->nxp;
    hp.has_z = v->has_z;
    hp.w64 = v->w64;
    hp.flags = v->flags;
    hp.has_bad = v->bad != nullptr;
    hp.has_done_list = v->done_list != nullptr;
    hp.pt_nq = v->pt_nq;
    hp.pt_rm = v->pt_rm;
    hp.pauli_compact = v->pt_nq <= 24 && v->pt_rm == 8;
    return hp;
}
// env.step() (one step per launch) of this handle keeps the tracked observation current itself
static bool dense_rides_in_step(const qg_vec *v) {
    if (!v->dense) return false;
    const plan::HandlePlan hp = plan_of(v);
    return plan::dense_in_kernel(hp, plan::step_kernel_of(hp, 1, false, false, v->maybe_nonsymplectic, (uint32_t)v->gates.size(), false));
}
static int dense_refresh(qg_vec *v, hipStream_t s);

int qg_vec_create(const qg_config *cfg, const qg_gate *gates, size_t n_gates, uint64_t batch, int device, qg_vec **out) {
    if (!cfg || !out || (!gates && n_gates)) return set_error(QG_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->num_qubits <= 0) return set_error(QG_ERR_INVALID, "num_qubits must be positive");
    if (cfg->max_depth < 0 || cfg->depth_slope < 0 || cfg->difficulty < 0)
        return set_error(QG_ERR_INVALID, "difficulty, depth_slope and max_depth are usize in the reference");
    if (batch == 0) return set_error(QG_ERR_INVALID, "batch must be positive");
    const uint32_t N = (uint32_t)cfg->num_qubits;
    for (size_t i = 0; i < n_gates; ++i) {
        const qg_gate &g = gates[i];
        if (g.kind < 0 || g.kind > QG_SWAP) return set_error(QG_ERR_INVALID, "gate %zu: unknown kind %d", i, g.kind);
        bool two = g.kind >= QG_CX;
        // the reference indexes rows by these qubits and would panic on the first step that uses them
        if (g.q0 < 0 || (uint32_t)g.q0 >= N || (two && (g.q1 < 0 || (uint32_t)g.q1 >= N)))
            return set_error(QG_ERR_INVALID, "gate %zu: qubit index out of range for %u qubits", i, N);
    }

    int ndev = qg_device_count();
    if (ndev <= 0) return set_error(QG_ERR_DEVICE, "no HIP device visible: libqgym has no CPU fallback");
    if (device < 0 || device >= ndev) return set_error(QG_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    qg::DeviceGuard guard(device);  // the caller's current device is restored on return
    if (guard.err != hipSuccess) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "cannot select device %d: %s", device, hipGetErrorString(guard.err));
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(QG_ERR_DEVICE, "device %d is %s; libqgym is built for gfx950 (MI355X) only", device, prop.gcnArchName);

    std::unique_ptr<qg_vec> v(new qg_vec());
    v->cfg = *cfg;
    v->gates.assign(gates, gates + n_gates);
    v->B = batch;
    v->N = N;
    v->device = device;
    v->difficulty = cfg->difficulty;
    v->coin_seed = 0x5EED0000C01Full;

    // layout, size class and behaviour flags: qgym_plan.hpp (the same function answers qg_plan_query without a GPU)
    plan::HandlePlan hp;
    const char *why = "";
    if (int rc = plan::handle_plan(*cfg, batch, hp, why)) return set_error(rc, "%s (num_qubits = %u)", why, N);
    v->layout = hp.layout;
    v->D = hp.D;
    v->nxp = hp.nxp;
    v->has_z = hp.has_z;
    v->w64 = hp.w64;
    v->flags = hp.flags;
    v->stride_bytes = hp.stride_bytes;
    v->state_bytes = hp.state_bytes;
    if (hp.layout == LAYOUT_PAULI) {
        v->rmax = hp.rmax;
        v->rmax_generate = hp.rmax_generate;
        v->cfg.max_rotations = hp.max_rotations;
        v->pt_nq = hp.pt_nq;
        v->pt_rm = hp.pt_rm;
    }
    const bool layers = v->flags & F_LAYERS;

    // gate table
    std::vector<GateEntry> table(std::max<size_t>(n_gates, 1));
    std::vector<uint32_t> descs(std::max<size_t>(n_gates, 1));
    const float w[4] = {cfg->w_n_cnots, cfg->w_n_layers_cnots, cfg->w_n_layers, cfg->w_n_gates};
    for (size_t i = 0; i < n_gates; ++i) {
        int dc, dg;
        gate_deltas(gates[i], N, dc, dg);
        table[i].ops = cfg->env_kind == QG_PAULI ? 0u
                       : v->layout == LAYOUT_TILE ? tile_ops(cfg->env_kind, gates[i], false)
                       : v->layout == LAYOUT_TILE64 ? tile_ops(cfg->env_kind, gates[i], true)
                                                  : gate_ops(cfg->env_kind, gates[i], N);
        table[i].penalty = table_penalty(w, dc, dg);
        descs[i] = make_desc((uint32_t)gates[i].kind, (uint32_t)gates[i].q0, (uint32_t)gates[i].q1);
    }

    qg_vec *p = v.get();
    auto fail = [&](int rc) {
        vec_free_buffers(p);
        return rc;
    };
#define HIP_TRY_V(expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (void)hipGetLastError();                                                               \
            return fail(set_error(QG_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)));  \
        }                                                                                          \
    } while (0)

    HIP_TRY_V(hipMalloc(&p->state, p->state_bytes));
    HIP_TRY_V(hipMemset(p->state, 0, p->state_bytes));
    HIP_TRY_V(hipMalloc(&p->depth, sizeof(int32_t) * batch));
    HIP_TRY_V(hipMalloc(&p->reward, sizeof(float) * batch));
    HIP_TRY_V(hipMalloc(&p->done, batch));
    HIP_TRY_V(hipMalloc(&p->success, batch));
    HIP_TRY_V(hipMalloc(&p->inverted, batch));
    // TILE / TILE64 layouts without add_inverts: the one-step kernel keeps `solved` as a per-env mask; PERMB: the number of entries with
    // state[i] != i; LFD: row masks of the state's and the inverse's region
    if (hp.has_bad)
        HIP_TRY_V(hipMalloc(&p->bad, (v->layout == LAYOUT_LFD ? 2 * sizeof(uint64_t) : v->layout == LAYOUT_TILE64 ? sizeof(uint64_t) : sizeof(uint32_t)) * batch));
    if (hp.has_done_list) {
        HIP_TRY_V(hipMalloc(&p->done_list, sizeof(uint32_t) * (batch + 2)));
        HIP_TRY_V(hipMemset(p->done_list + batch, 0, 2 * sizeof(uint32_t)));
    }
    if (hp.has_done_list && (v->layout == LAYOUT_TILE || v->layout == LAYOUT_TILE64)) {  // the init kernel zeroes the idle list's length instead of taking reader tickets on its own
        HIP_TRY_V(hipMalloc(&p->done_list_spare, sizeof(uint32_t) * (batch + 2)));
        HIP_TRY_V(hipMemset(p->done_list_spare + batch, 0, 2 * sizeof(uint32_t)));
    }
    if (plan::reset_step_fusable(hp)) {  // qg_vec_reset_done_step in one launch: the second list
        HIP_TRY_V(hipMalloc(&p->done_list_alt, sizeof(uint32_t) * (batch + 2)));
        HIP_TRY_V(hipMemset(p->done_list_alt + batch, 0, 2 * sizeof(uint32_t)));
    }
    if (hp.has_done_list && (v->layout == LAYOUT_TILE || v->layout == LAYOUT_TILE64 || v->layout == LAYOUT_PAULI)) {  // the finished envs of a step as one bit each (qm_step1 / qm_inv2 / q64_step1 / q64_inv2 / ptile_step1c <LIST>)
        HIP_TRY_V(hipHostMalloc((void **)&p->count_seen, sizeof(uint32_t), hipHostMallocMapped));
        *p->count_seen = 0xFFFFFFFFu;
        if (v->layout != LAYOUT_TILE) {
            HIP_TRY_V(hipMalloc(&p->mask_count, 2 * sizeof(uint32_t)));
            HIP_TRY_V(hipMemset(p->mask_count, 0, 2 * sizeof(uint32_t)));
        }
        const size_t mask_bytes = 10 * 4 * ((batch + 255) / 256) + 8;  // device_common.hpp done_mask_bytes: a word per wave of the step grid (whole workgroups of 256 envs), then a count byte per 32 envs
        for (auto &m : p->done_mask) {
            HIP_TRY_V(hipMalloc(&m, mask_bytes));
            HIP_TRY_V(hipMemset(m, 0, mask_bytes));
        }
    }
    HIP_TRY_V(hipMalloc(&p->error, sizeof(uint32_t) * batch));
    HIP_TRY_V(hipMalloc(&p->sol_len, sizeof(int32_t) * 2 * batch));
    if (cfg->track_solution) {
        // Clifford/LF/Permutation push one entry per step; PauliEnv one per step plus one per removed rotation
        p->sol_cap = (uint32_t)std::max(cfg->max_depth, 1);
        if (v->layout == LAYOUT_PAULI) p->sol_cap += p->rmax;
        HIP_TRY_V(hipMalloc(&p->sol, sizeof(uint32_t) * (size_t)p->sol_cap * batch));
    }
    if (layers) {
        p->layers_len = 2 * N + 2;
        HIP_TRY_V(hipMalloc(&p->layers, sizeof(int32_t) * (size_t)p->layers_len * (((size_t)batch + 63) & ~(size_t)63)));  // tiles of 64 envs (layer_rec)
    }
    HIP_TRY_V(hipMalloc(&p->d_gates, sizeof(GateEntry) * table.size()));
    HIP_TRY_V(hipMalloc(&p->d_descs, sizeof(uint32_t) * descs.size()));
    HIP_TRY_V(hipMemcpy(p->d_gates, table.data(), sizeof(GateEntry) * table.size(), hipMemcpyHostToDevice));
    HIP_TRY_V(hipMemcpy(p->d_descs, descs.data(), sizeof(uint32_t) * descs.size(), hipMemcpyHostToDevice));
    if (v->layout == LAYOUT_TILE || v->layout == LAYOUT_TILE64) {  // the same gates as <= 2 row operations on tile slots (clifford.rs:89-133)
        std::vector<uint32_t> rowops(table.size(), 0u);
        auto slot = [&](uint32_t row) { return v->has_z ? (row < N ? 2 * row : 2 * (row - N) + 1) : row; };
        for (size_t i = 0; i < n_gates; ++i) {
            const uint32_t ops = gate_ops(cfg->env_kind, gates[i], N);
            uint32_t out = 0;
            for (int k = 0; k < 2; ++k) {
                const uint32_t op = (ops >> (14 * k)) & 0x3FFFu, type = (op >> 12) & 3u;
                if (type != OP_NONE) out |= make_op(type, slot(op & 63u), slot((op >> 6) & 63u)) << (14 * k);
            }
            rowops[i] = out;
        }
        HIP_TRY_V(hipMalloc(&p->d_rowops, sizeof(uint32_t) * rowops.size()));
        HIP_TRY_V(hipMemcpy(p->d_rowops, rowops.data(), sizeof(uint32_t) * rowops.size(), hipMemcpyHostToDevice));
    }
    if (v->layout == LAYOUT_PAULI) {
        int rc = pauli_alloc(p);
        if (rc) return fail(rc);
    }

    // observe_dense of the 64-bit-row / lane-group / PauliEnv layouts goes through row words in the handle's scratch buffer: sized here, so
    // that the call never allocates (it may be made inside a stream capture)
    if (v->layout == LAYOUT_TILE64 || v->layout == LAYOUT_LFD || v->layout == LAYOUT_PAULI) {
        const size_t word = (v->layout == LAYOUT_LFD && !v->w64) ? 4 : 8;
        int rc = ensure_scratch(p, (size_t)batch * v->D * word);
        if (rc) return fail(rc);
    }
    // constructor state: identity, depth 1, success, reward 1.0 (clifford.rs:214-245)
    if (v->layout == LAYOUT_PAULI) {
        int rc = pauli_init_identity(p, nullptr);
        if (rc) return fail(rc);
    } else {
        InitArgs ia;
        fill_init_args(p, ia);
        ia.mode = 0;
        ia.depth_value = 1;
        HIP_TRY_V(launch_init(p, ia, nullptr));
    }
    HIP_TRY_V(hipDeviceSynchronize());
#undef HIP_TRY_V
    *out = v.release();
    return QG_OK;
}

void qg_vec_destroy(qg_vec *v) {
    if (!v) return;
    {
        qg::DeviceGuard guard(v->device);
        (void)hipDeviceSynchronize();
        vec_free_buffers(v);
    }
    delete v;
}

int qg_vec_get_info(const qg_vec *v, qg_vec_info *o) {
    if (!v || !o) return set_error(QG_ERR_INVALID, "null argument");
    memset(o, 0, sizeof *o);
    o->env_kind = v->cfg.env_kind;
    o->num_qubits = (int32_t)v->N;
    o->num_actions = (int32_t)v->gates.size();
    switch (v->cfg.env_kind) {
    case QG_CLIFFORD: o->obs_rows = o->obs_cols = (int32_t)(2 * v->N); break;  // clifford.rs:291-294
    case QG_PAULI:                                                             // pauli.rs:505-507
        o->obs_rows = (int32_t)(2 * v->N);
        o->obs_cols = (int32_t)(2 * v->N + std::max(v->cfg.max_rotations, 1));
        break;
    default: o->obs_rows = o->obs_cols = (int32_t)v->N; break;
    }
    o->device = v->device;
    o->batch = v->B;
    o->packed_word_bytes = (v->layout == LAYOUT_PERM || v->layout == LAYOUT_PERMB) ? 1 : ((v->layout == LAYOUT_TILE64 || v->layout == LAYOUT_PAULI || (v->layout == LAYOUT_LFD && v->w64)) ? 8 : 4);
    o->packed_words_per_env = v->D;
    o->packed_env_stride_bytes = v->stride_bytes;
    o->state_dev = v->state;
    o->reward_dev = v->reward;
    o->done_dev = v->done;
    o->success_dev = v->success;
    o->depth_dev = v->depth;
    o->error_dev = v->error;
    return QG_OK;
}

static void drop_graphs(qg_vec *v) {
    for (auto &g : v->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    v->graphs.clear();
}

int qg_vec_bind_outputs(qg_vec *v, float *reward_dev, uint8_t *done_dev, uint8_t *success_dev, int32_t *depth_dev) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    HIP_TRY(hipDeviceSynchronize());
    drop_graphs(v);  // cached graphs have the old pointers baked in
    auto rebind = [&](auto *&cur, auto *ext, bool &own, size_t bytes) -> int {
        if (!ext || ext == cur) return QG_OK;
        HIP_TRY(hipMemcpy(ext, cur, bytes, hipMemcpyDeviceToDevice));
        if (own) HIP_TRY(hipFree(cur));
        cur = ext;
        own = false;
        return QG_OK;
    };
    int rc;
    if ((rc = rebind(v->reward, reward_dev, v->own_reward, sizeof(float) * v->B))) return rc;
    if ((rc = rebind(v->done, done_dev, v->own_done, v->B))) return rc;
    if ((rc = rebind(v->success, success_dev, v->own_success, v->B))) return rc;
    if ((rc = rebind(v->depth, depth_dev, v->own_depth, sizeof(int32_t) * v->B))) return rc;
    return QG_OK;
}

int qg_vec_set_difficulty(qg_vec *v, int64_t d) {
    if (!v || d < 0) return set_error(QG_ERR_INVALID, "bad difficulty");
    v->difficulty = d;
    return QG_OK;
}
int64_t qg_vec_get_difficulty(const qg_vec *v) { return v ? v->difficulty : -1; }

static size_t format_elem_bytes(const qg_vec *v, int format) {
    if (format == QG_FMT_I64) return 8;
    if (format == QG_FMT_U8) return 1;
    return (v->layout == LAYOUT_PERM || v->layout == LAYOUT_PERMB) ? 1 : ((v->layout == LAYOUT_TILE64 || v->layout == LAYOUT_PAULI || (v->layout == LAYOUT_LFD && v->w64)) ? 8 : 4);
}
static size_t format_min_elems(const qg_vec *v, int format) {
    if (v->layout == LAYOUT_PERM || v->layout == LAYOUT_PERMB) return v->N;
    if (format == QG_FMT_PACKED) return v->D;
    return (size_t)v->D * v->D;
}

int qg_vec_set_state(qg_vec *v, const void *states, int format, size_t stride, int on_device, void *stream) {
    if (!v || !states) return set_error(QG_ERR_INVALID, "null argument");
    if (format < QG_FMT_I64 || format > QG_FMT_PACKED) return set_error(QG_ERR_INVALID, "unknown state format %d", format);
    QG_ON_DEVICE(v);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = drop_done_list(v, s)) return rc;
    if (v->layout == LAYOUT_PAULI) return pauli_set_state(v, states, format, stride, on_device, s);
    if (stride < format_min_elems(v, format))
        return set_error(QG_ERR_INVALID, "set_state: %zu elements per env, need %zu (the reference would index out of bounds)",
                         stride, format_min_elems(v, format));
    const void *src = states;
    // entry formats (the trait's Vec<i64>, dense bytes) of the bit-matrix layouts: the flat entry stream becomes a bit stream first (64 entries
    // per wave instruction, coalesced), the init kernel cuts its row words out of it -- no thread walks rows of 8-byte entries
    // (a handful of envs -- the scalar qg_env_* handles are batches of one -- keep the single launch)
    const bool as_bits = format != QG_FMT_PACKED && plan::entry_formats_stream(v->layout, v->B);
    const size_t in_bytes = format_elem_bytes(v, format) * stride * v->B;
    const size_t staged = on_device ? 0 : (in_bytes + 15) & ~(size_t)15;
    const size_t stream_words = as_bits ? (stride * v->B + 63) / 64 + 2 : 0;  // + the word bits_window may read past the end
    if (staged + stream_words) {
        int rc = ensure_scratch(v, staged + stream_words * sizeof(uint64_t));
        if (rc) return rc;
    }
    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(v->scratch, states, in_bytes, hipMemcpyHostToDevice, s));
        src = v->scratch;
    }
    InitArgs ia;
    fill_init_args(v, ia);
    ia.mode = 1;
    ia.src = src;
    ia.src_stride = stride;
    ia.format = (uint32_t)format;
    if (as_bits) {
        uint64_t *stream = reinterpret_cast<uint64_t *>((char *)v->scratch + staged);
        HIP_TRY(hipMemsetAsync(stream + stream_words - 2, 0, 2 * sizeof(uint64_t), s));
        HIP_TRY(pack_bitstream(src, (int)format_elem_bytes(v, format), (uint64_t)stride * v->B, stream, s));
        ia.src = stream;
        ia.format = QG_FMT_BITS;
    }
    ia.depth_value = v->cfg.max_depth;  // clifford.rs:302
    if (ia.check_symplectic) {  // the init kernel reports whether any installed matrix is not symplectic
        if (!v->d_nonsymp) HIP_TRY(hipMalloc(&v->d_nonsymp, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(v->d_nonsymp, 0, sizeof(uint32_t), s));
        ia.nonsymp_flag = v->d_nonsymp;
    }
    HIP_TRY(launch_init(v, ia, s));
    if (v->dense)
        if (int rc = dense_refresh(v, s)) return rc;
    if (ia.check_symplectic) {
        uint32_t flag = 0;
        HIP_TRY(hipMemcpyAsync(&flag, v->d_nonsymp, sizeof flag, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        v->maybe_nonsymplectic = flag != 0;  // picks the step-kernel variant with the Gauss-Jordan inversion
    } else if (!on_device) {
        HIP_TRY(hipStreamSynchronize(s));  // the host buffer may be reused by the caller
    }
    return QG_OK;
}

static hipError_t launch_export(const qg_vec *v, const ObsArgs &a_in, hipStream_t s) {
    ObsArgs a = a_in;
    a.kclk = kernel_clock_slot(v);
    a.kclk_waves = v->kclk_waves;
    switch (v->layout) {
    case LAYOUT_LFD: return lfd_export(a, v->w64, v->nxp, v->inverted, s);
    case LAYOUT_LF8: return lf8_export(a, s);
    case LAYOUT_PERM: return perm_export(a, s);
    case LAYOUT_PERMB: return permb_export(a, v->nxp, s);
    case LAYOUT_PAULI: return pauli_export(v, a, s);
    case LAYOUT_TILE: return qm_export(a, v->nxp, v->has_z, s);
    case LAYOUT_TILE64: return q64_export(a, v->nxp, v->has_z, s);
    default: return hipErrorInvalidValue;
    }
}

static void fill_obs_args(const qg_vec *v, ObsArgs &a, void *out, int format, size_t stride) {
    memset(&a, 0, sizeof a);
    a.state = v->state;
    a.out = out;
    a.B = v->B;
    a.out_stride = stride;
    a.D = v->D;
    a.N = v->N;
    a.log2L = v->log2L;
    a.format = (uint32_t)format;
    qg_vec_info info;
    qg_vec_get_info(v, &info);
    a.obs_rows = (uint32_t)info.obs_rows;
    a.obs_cols = (uint32_t)info.obs_cols;
}

int qg_vec_get_state(qg_vec *v, void *out, int format, size_t stride, int on_device, void *stream) {
    if (!v || !out) return set_error(QG_ERR_INVALID, "null argument");
    if (format < QG_FMT_I64 || format > QG_FMT_PACKED) return set_error(QG_ERR_INVALID, "unknown state format %d", format);
    if (stride < format_min_elems(v, format)) return set_error(QG_ERR_INVALID, "get_state: stride too small");
    QG_ON_DEVICE(v);
    hipStream_t s = (hipStream_t)stream;
    void *dst = out;
    size_t bytes = format_elem_bytes(v, format) * stride * v->B;
    // the trait's Vec<i64> of the bit-matrix layouts: the row words first (8 MiB for CliffordEnv 16q x 65 536), then one expansion kernel whose
    // stores are wave-contiguous -- 537 MB at streaming rate (the export kernels' row-per-thread, entry-by-entry stores reached 0.6 TB/s)
    qg_vec_info info;
    qg_vec_get_info(v, &info);
    const bool two_stage = format == QG_FMT_I64 && plan::entry_formats_stream(v->layout, v->B) && stride == (size_t)v->D * v->D && !((uintptr_t)out & 15u) &&
                           info.packed_words_per_env == v->D;
    const size_t staged = on_device ? 0 : (bytes + 15) & ~(size_t)15;
    const size_t words_bytes = two_stage ? (size_t)v->B * v->D * info.packed_word_bytes : 0;
    if (staged + words_bytes) {
        int rc = ensure_scratch(v, staged + words_bytes);
        if (rc) return rc;
    }
    if (!on_device) dst = v->scratch;
    ObsArgs oa;
    if (two_stage) {
        void *words = (char *)v->scratch + staged;
        fill_obs_args(v, oa, words, QG_FMT_PACKED, v->D);
        HIP_TRY(launch_export(v, oa, s));
        HIP_TRY(expand_rows_i64(words, (int)info.packed_word_bytes, v->B * (uint64_t)v->D, v->D, reinterpret_cast<int64_t *>(dst), s));
    } else {
        fill_obs_args(v, oa, dst, format, stride);
        if (v->layout == LAYOUT_PAULI) {  // tableau only, square
            oa.obs_cols = oa.obs_rows;
        }
        HIP_TRY(launch_export(v, oa, s));
    }
    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return QG_OK;
}

static int do_reset(qg_vec *v, const int32_t *actions_dev, size_t n_draws, uint64_t seed, hipStream_t s, bool only_done = false) {
    QG_ON_DEVICE(v);
    const bool trusted = done_list_session(v, s);
    if (!only_done)
        if (int rc = drop_done_list(v, s)) return rc;
    if (only_done && v->layout == LAYOUT_PAULI) {
        const bool from_mask = trusted && v->done_list_fresh && v->mask_fresh;  // the step before (ptile_step1c_kernel<LIST>) left its finishers as bits
        v->done_list_fresh = v->mask_fresh = false;
        v->auto_list = true;  // (from now on single steps leave their finishers themselves)
        return ptile_reset_seeded(v, seed, true, s, from_mask);
    }
    if (v->layout == LAYOUT_PAULI) {
        if (actions_dev) return set_error(QG_ERR_UNSUPPORTED, "PauliEnv reset draws a whole target, not `difficulty` actions: use qg_vec_reset(seed) or qg_vec_pauli_reset_from");
        return pauli_reset_seeded(v, seed, s);
    }
    if (v->gates.empty() && n_draws)  // Uniform::new(0, 0) panics in the reference
        return set_error(QG_ERR_PANIC, "reset with an empty gateset (the reference panics in Uniform::new(0, 0))");
    InitArgs ia;
    fill_init_args(v, ia);
    ia.mode = 2;
    ia.actions = actions_dev;
    ia.n_draws = (uint32_t)n_draws;
    ia.seed = seed;
    ia.only_done = only_done ? 1u : 0u;
    if (only_done && v->done_list) {
        // few, scattered finished envs: pack their indices first so that the scramble runs in full waves
        // instead of in every wave that holds one finished env (the sampling + step kernel has already done it: done_list_fresh)
        const bool left_by_step = trusted && v->done_list_fresh;  // the step before recorded its finishers itself (a list, or TILE: bits + a list)
        if (!left_by_step) {
            HIP_TRY(compact_done(v->done, v->B, v->done_list, v->done_list + v->B, s));
            v->mask_fresh = false;
        }
        v->done_list_fresh = false;
        v->list_zero_known = true;  // the reset kernel is the list's consumer: it zeroes the length (list_count_take)
        v->auto_list = true;
        ia.list = v->done_list;
        ia.list_count = v->done_list + v->B;
        if ((v->layout == LAYOUT_TILE || v->layout == LAYOUT_TILE64) && v->count_seen) {
            ia.count_out = v->count_seen;
            ia.tree_grid = reset_tree_grid(v, plan::tree_grid(v->B));
        }
        if (left_by_step && v->mask_fresh && v->done_mask[0]) {  // TILE: the step before left its finishers as bits (the list holds what the fused launch added, if anything)
            ia.mask = v->done_mask[v->mask_cur];
            ia.mask_words = (uint32_t)(4 * ((v->B + 255) / 256));
            ia.mask_epoch = v->mask_epoch[v->mask_cur];
            ia.count_pub = v->mask_count;
        }
        v->mask_fresh = false;
        ia.coop = plan::reset_coop_allowed(actions_dev != nullptr, v->B, v->d_rowops != nullptr) ? 1u : 0u;
        ia.flags_current = 1u;  // (this launch is the reset alone)
        if (v->layout == LAYOUT_TILE) ia.dense = v->dense;  // the listed envs' dense observations are rewritten by the reset itself
        if (v->done_list_spare) ia.zero_count = v->done_list_spare + v->B;
    }
    if (!only_done) v->maybe_nonsymplectic = false;  // identity + gates: every env is symplectic again
    int64_t d = (int64_t)v->cfg.depth_slope * v->difficulty;  // clifford.rs:317
    ia.depth_value = (int32_t)std::min<int64_t>(d, v->cfg.max_depth);
    HIP_TRY(launch_init(v, ia, s));
    if (ia.zero_count) std::swap(v->done_list, v->done_list_spare);  // the list just zeroed is the one the next step appends to; the consumed one idles
    if (v->dense && !ia.dense) return dense_refresh(v, s);
    return QG_OK;
}

int qg_vec_set_clock(qg_vec *v, const uint64_t *clock_dev) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    HIP_TRY(hipDeviceSynchronize());
    v->clock_dev = clock_dev;
    // cached rollout graphs bake the old pointer in
    for (auto &g : v->graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    v->graphs.clear();
    return QG_OK;
}

int qg_vec_set_kernel_clock(qg_vec *v, uint64_t *slots_dev, size_t n_slots, uint32_t waves_per_slot) {
    if (!v || (n_slots && (!slots_dev || !waves_per_slot))) return set_error(QG_ERR_INVALID, "null argument");
    if ((uintptr_t)slots_dev & 15u) return set_error(QG_ERR_INVALID, "kernel clock slots must be 16-byte aligned");
    drop_graphs(v);  // cached rollout graphs carry the previous slots (or none) in their kernel arguments
    v->kclk = n_slots ? reinterpret_cast<unsigned long long *>(slots_dev) : nullptr;
    v->kclk_cap = n_slots;
    v->kclk_waves = n_slots ? waves_per_slot : 0u;
    v->kclk_next = 0;
    return QG_OK;
}

int qg_kernel_clock_rate_khz(int device) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "cannot read the wall clock rate of device %d", device);
    }
    return khz;
}

int qg_stream_wait_stream(void *waiter, void *producer) {
    // one pooled event per thread and device (an event belongs to the device it was created on): device-scope only
    // (no system fence), no timing.  Both streams must belong to the calling thread's current device.
    static thread_local hipEvent_t evs[64] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return set_error(QG_ERR_INVALID, "device index %d out of range", dev);
    if (!evs[dev]) HIP_TRY(hipEventCreateWithFlags(&evs[dev], hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_TRY(hipEventRecord(evs[dev], (hipStream_t)producer));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)waiter, evs[dev], 0));
    return QG_OK;
}

int qg_vec_set_counters(qg_vec *v, uint64_t step_index, uint64_t observe_index) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    v->step_index = step_index;
    v->observe_counter = observe_index;
    return QG_OK;
}

int qg_vec_set_seed(qg_vec *v, uint64_t seed) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    if (seed != v->coin_seed) {
        QG_ON_DEVICE(v);
        HIP_TRY(hipDeviceSynchronize());
        drop_graphs(v);  // the seed is a baked-in kernel argument of captured launches
        v->coin_seed = seed;
    }
    return QG_OK;
}

int qg_vec_set_env_base(qg_vec *v, uint64_t first_env) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    v->env_base = first_env;  // cached graphs are keyed by it
    return QG_OK;
}
uint64_t qg_vec_get_env_base(const qg_vec *v) { return v ? v->env_base : 0; }

int qg_vec_reset(qg_vec *v, uint64_t seed, void *stream) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    return do_reset(v, nullptr, (size_t)v->difficulty, seed, (hipStream_t)stream);
}
int qg_vec_reset_done(qg_vec *v, uint64_t seed, void *stream) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    return do_reset(v, nullptr, (size_t)v->difficulty, seed, (hipStream_t)stream, true);
}

int qg_vec_reset_with(qg_vec *v, const int32_t *actions_dev, size_t n_draws, void *stream) {
    if (!v || (!actions_dev && n_draws)) return set_error(QG_ERR_INVALID, "null argument");
    if ((int64_t)n_draws != v->difficulty)
        return set_error(QG_ERR_INVALID, "reset_with: need exactly `difficulty` (%lld) draws per env, got %zu",
                         (long long)v->difficulty, n_draws);
    return do_reset(v, actions_dev, n_draws, 0, (hipStream_t)stream);
}

// A handle on which qg_vec_reset_done is in use: a single step of the TILE / TILE64 one-step kernels (qm_step1, qm_inv2, q64_step1, q64_inv2)
// leaves the list of the envs it finished itself, and the reset that follows needs no compaction launch.
static bool step_leaves_done_list(const qg_vec *v, StepArgs &a) {
    const bool tile32 = v->layout == LAYOUT_TILE, tile64 = v->layout == LAYOUT_TILE64;  // qm_step1 / qm_inv2 (N <= 16), q64_step1 / q64_inv2
    const bool pauli = v->layout == LAYOUT_PAULI && v->done_mask[0] && v->pt_nq <= 24 && v->pt_rm == 8 && v->B > QG_COMPACT_MIN_ENVS;  // ptile_step1c_kernel (compact layout)
    const bool lists = v->auto_list && v->done_list &&
                       (pauli || ((tile32 || tile64) && ((v->flags & F_INVERTS) ? (v->has_z && (tile64 || v->nxp <= 16) && !v->maybe_nonsymplectic) : v->bad != nullptr)));
    if (lists) {
        a.flags |= F_DONE_LIST;
        a.done_list = v->done_list;
        a.done_count = v->done_list + v->B;
        if (v->done_mask[0]) {  // one bit per env instead of an append (the list stays empty)
            a.done_mask = v->done_mask[v->mask_cur ^ 1];
            a.done_epoch = mask_epoch_of(v);
        }
    }
    return lists;
}

int qg_vec_step(qg_vec *v, const void *actions_dev, int action_dtype, const uint8_t *coins_dev, void *stream) {
    if (!v || !actions_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (action_dtype != QG_ACT_I32 && action_dtype != QG_ACT_I64) return set_error(QG_ERR_INVALID, "bad action dtype");
    QG_ON_DEVICE(v);
    const bool trusted = done_list_session(v, (hipStream_t)stream);
    if (v->done_list_fresh) v->auto_list = false;  // the last list was never consumed: this caller steps without qg_vec_reset_done
    StepArgs a;
    fill_step_args(v, a);
    const bool lists = trusted && step_leaves_done_list(v, a);
    if (int rc = lists ? done_list_before_append(v, (hipStream_t)stream) : drop_done_list(v, (hipStream_t)stream)) return rc;
    a.actions = actions_dev;
    a.coins = coins_dev;
    if (action_dtype == QG_ACT_I64) a.flags |= F_ACT64;
    if (dense_rides_in_step(v)) a.dense = v->dense;
    HIP_TRY(launch_step(v, a, (hipStream_t)stream));
    v->step_index += 1;
    if (lists) {
        done_list_appended(v, true);
        step_wrote_mask(v, a, true);
    }
    if (v->dense && !a.dense) return dense_refresh(v, (hipStream_t)stream);
    return QG_OK;
}

// the device's view of `bytes` of pinned, mapped host memory; nullptr for pageable memory, for anything the runtime does not know, and for a
// mapping that does not cover the whole range
static void *mapped_view(const void *host_ptr, size_t bytes) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)at.devicePointer) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    const uintptr_t lo = (uintptr_t)at.devicePointer, end = (uintptr_t)base + size;
    return (lo + bytes <= end) ? at.devicePointer : nullptr;
}

int qg_vec_step_host(qg_vec *v, const void *actions_host, int action_dtype, const uint8_t *coins_host, float *rewards_host, uint8_t *dones_host,
                     uint8_t *success_host, void *stream) {
    if (!v || !actions_host) return set_error(QG_ERR_INVALID, "null argument");
    if (action_dtype != QG_ACT_I32 && action_dtype != QG_ACT_I64) return set_error(QG_ERR_INVALID, "bad action dtype");
    QG_ON_DEVICE(v);
    hipStream_t s = (hipStream_t)stream;
    {   // pinned, device-mapped buffers (hipHostMalloc / hipHostRegister): the step reads the actions where they are and one small kernel
        // writes the outputs where they go -- two launches, no copy engine (four copies of <= 256 KiB cost ~13 us of latency each)
        const size_t act_bytes_m = (action_dtype == QG_ACT_I64 ? 8 : 4) * v->B;
        void *act_m = mapped_view(actions_host, act_bytes_m), *coin_m = coins_host ? mapped_view(coins_host, v->B) : nullptr;
        void *rew_m = rewards_host ? mapped_view(rewards_host, sizeof(float) * v->B) : nullptr, *done_m = dones_host ? mapped_view(dones_host, v->B) : nullptr,
             *suc_m = success_host ? mapped_view(success_host, v->B) : nullptr;
        // step_outputs_kernel moves 16 bytes of rewards and 4 bytes of flags per thread: destinations AND sources (the handle's output arrays
        // may be bound to caller memory of any alignment, qg_vec_bind_outputs) must allow it, else the copies below do the job
        const bool aligned = !(((uintptr_t)rew_m | (uintptr_t)done_m | (uintptr_t)suc_m) & 15u) && !((uintptr_t)v->reward & 15u) &&
                             !(((uintptr_t)v->done | (uintptr_t)v->success) & 3u);
        if (act_m && (!coins_host || coin_m) && (!rewards_host || rew_m) && (!dones_host || done_m) && (!success_host || suc_m) && aligned) {
            if (int rc = qg_vec_step(v, act_m, action_dtype, (const uint8_t *)coin_m, stream)) return rc;
            HIP_TRY(step_outputs(v->reward, v->done, v->success, (float *)rew_m, (uint8_t *)done_m, (uint8_t *)suc_m, v->B, s));
            return QG_OK;
        }
    }
    if (!v->host_in) HIP_TRY(hipMalloc(&v->host_in, 9 * v->B));
    const size_t act_bytes = (action_dtype == QG_ACT_I64 ? 8 : 4) * v->B;
    uint8_t *coins_dev = coins_host ? (uint8_t *)v->host_in + 8 * v->B : nullptr;
    HIP_TRY(hipMemcpyAsync(v->host_in, actions_host, act_bytes, hipMemcpyHostToDevice, s));
    if (coins_host) HIP_TRY(hipMemcpyAsync(coins_dev, coins_host, v->B, hipMemcpyHostToDevice, s));
    if (int rc = qg_vec_step(v, v->host_in, action_dtype, coins_dev, stream)) return rc;
    if (rewards_host) HIP_TRY(hipMemcpyAsync(rewards_host, v->reward, sizeof(float) * v->B, hipMemcpyDeviceToHost, s));
    if (dones_host) HIP_TRY(hipMemcpyAsync(dones_host, v->done, v->B, hipMemcpyDeviceToHost, s));
    if (success_host) HIP_TRY(hipMemcpyAsync(success_host, v->success, v->B, hipMemcpyDeviceToHost, s));
    return QG_OK;
}

static int rollout_impl(qg_vec *v, const void *actions_dev, int action_dtype, size_t T, size_t period, const uint8_t *coins_dev,
                        float *rewards_dev, uint8_t *dones_dev, int fused, void *stream);

int qg_vec_reset_done_step(qg_vec *v, uint64_t reset_seed, const void *actions_dev, int action_dtype, const uint8_t *coins_dev, float *rewards_dev,
                           uint8_t *dones_dev, void *stream) {
    if (!v || !actions_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (action_dtype != QG_ACT_I32 && action_dtype != QG_ACT_I64) return set_error(QG_ERR_INVALID, "bad action dtype");
    QG_ON_DEVICE(v);
    hipStream_t s = (hipStream_t)stream;
    const bool trusted = done_list_session(v, s);
    // one launch when the list of finished envs and their is_final flags were left by this handle's own previous step (same session) and the
    // reset would take the list path with counter-RNG draws; otherwise the two calls, whose step leaves both for the next time
    // (add_inverts: the two-lanes-per-env launch has no Gauss-Jordan and no tracked-observation form)
    const bool inverts = v->flags & F_INVERTS;
    // ... and where the one launch is the faster form (qgym_plan.hpp reset_step_pays: from the configuration)
    const bool pays = plan::reset_step_pays(plan_of(v), v->difficulty, std::min<int64_t>((int64_t)v->cfg.depth_slope * v->difficulty, v->cfg.max_depth));
    const bool fuse = v->done_list_alt && v->done_mask[0] && trusted && v->done_list_fresh && v->mask_fresh && v->auto_list && !v->gates.empty() &&
                      plan::reset_coop_allowed(false, v->B, v->d_rowops != nullptr) && plan::reset_step_fuses(plan_of(v)) &&
                      !(inverts && (v->maybe_nonsymplectic || v->dense)) && pays;
    if (plan::reset_step_in_word_kernel(plan_of(v), v->gates.size())) {
        // one uint64 per env: no list -- every wave tests its envs' is_final flags, resets the finished ones (16 lanes each) and steps all of them
        InitArgs ia;
        fill_init_args(v, ia);
        ia.mode = 2;
        ia.n_draws = (uint32_t)v->difficulty;
        ia.seed = reset_seed;
        ia.only_done = 1u;
        ia.depth_value = (int32_t)std::min<int64_t>((int64_t)v->cfg.depth_slope * v->difficulty, v->cfg.max_depth);  // linear_function.rs:296
        StepArgs a;
        fill_step_args(v, a);
        a.actions = actions_dev;
        a.coins = coins_dev;
        a.rewards_seq = rewards_dev;
        a.dones_seq = dones_dev;
        if (action_dtype == QG_ACT_I64) a.flags |= F_ACT64;
        a.kclk = kernel_clock_slot(v);
        a.kclk_waves = v->kclk_waves;
        HIP_TRY(word_reset_step(ia, a, v->layout == LAYOUT_PERM, s));
        v->step_index += 1;
        return QG_OK;
    }
    if (!fuse) {
        if (int rc = qg_vec_reset_done(v, reset_seed, stream)) return rc;
        return rollout_impl(v, actions_dev, action_dtype, 1, 1, coins_dev, rewards_dev, dones_dev, 0, stream);
    }
    InitArgs ia;
    fill_init_args(v, ia);
    ia.mode = 2;
    ia.n_draws = (uint32_t)v->difficulty;
    ia.seed = reset_seed;
    ia.only_done = 1u;
    ia.list = v->done_list;
    ia.list_count = v->done_list + v->B;
    ia.zero_count = v->done_list_spare + v->B;
    ia.count_out = v->count_seen;
    ia.tree_grid = reset_tree_grid(v, plan::tree_grid(v->B));
    ia.mask = v->done_mask[v->mask_cur];  // the finishers of the step before: the reset's work, and the step workgroups' "not mine" test
    ia.mask_words = (uint32_t)(4 * ((v->B + 255) / 256));
    ia.mask_epoch = v->mask_epoch[v->mask_cur];
    ia.coop = 1u;
    ia.dense = v->dense;
    ia.depth_value = (int32_t)std::min<int64_t>((int64_t)v->cfg.depth_slope * v->difficulty, v->cfg.max_depth);  // clifford.rs:317
    StepArgs a;
    fill_step_args(v, a);
    a.actions = actions_dev;
    a.coins = coins_dev;
    a.rewards_seq = rewards_dev;
    a.dones_seq = dones_dev;
    if (action_dtype == QG_ACT_I64) a.flags |= F_ACT64;
    a.flags |= F_DONE_LIST;  // the envs that finish in this step go to the OTHER mask (a reset env that is final again after its first step: to the OTHER list)
    a.done_list = v->done_list_alt;
    a.done_count = v->done_list_alt + v->B;
    a.done_mask = v->done_mask[v->mask_cur ^ 1];
    a.done_epoch = mask_epoch_of(v);
    if (!v->alt_zero_known) HIP_TRY(hipMemsetAsync(v->done_list_alt + v->B, 0, 2 * sizeof(uint32_t), s));
    if (dense_rides_in_step(v)) a.dense = v->dense;
    a.kclk = kernel_clock_slot(v);
    a.kclk_waves = v->kclk_waves;
    HIP_TRY(v->layout == LAYOUT_TILE64 ? q64_reset_step(ia, a, v->nxp, v->has_z, s) : qm_reset_step(ia, a, v->nxp, v->has_z, s));
    v->step_index += 1;
    // the list just appended to is the current one, the idle list (zeroed by this launch) is the next launch's target, the one just consumed idles
    std::swap(v->done_list, v->done_list_alt);   // (current, alt, spare) <- (alt, spare, current)
    std::swap(v->done_list_alt, v->done_list_spare);
    v->mask_cur ^= 1;
    v->mask_epoch[v->mask_cur] = a.done_epoch;
    v->done_list_fresh = true;
    v->mask_fresh = true;
    v->list_zero_known = false;
    v->alt_zero_known = true;
    if (v->dense && !a.dense) return dense_refresh(v, s);
    return QG_OK;
}

int qg_vec_rollout(qg_vec *v, const void *actions_dev, int action_dtype, size_t T, const uint8_t *coins_dev,
                   float *rewards_dev, uint8_t *dones_dev, int fused, void *stream) {
    return rollout_impl(v, actions_dev, action_dtype, T, T, coins_dev, rewards_dev, dones_dev, fused, stream);
}

int qg_vec_rollout_ring(qg_vec *v, const void *actions_dev, int action_dtype, size_t T, size_t period, void *stream) {
    if (period == 0) return set_error(QG_ERR_INVALID, "period must be positive");
    return rollout_impl(v, actions_dev, action_dtype, T, period, nullptr, nullptr, nullptr, 0, stream);
}

static int rollout_impl(qg_vec *v, const void *actions_dev, int action_dtype, size_t T, size_t period, const uint8_t *coins_dev,
                        float *rewards_dev, uint8_t *dones_dev, int fused, void *stream) {
    if (!v || !actions_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (action_dtype != QG_ACT_I32 && action_dtype != QG_ACT_I64) return set_error(QG_ERR_INVALID, "bad action dtype");
    if (T == 0) return QG_OK;
    if (T > 0x7fffffffu) return set_error(QG_ERR_INVALID, "too many steps");
    QG_ON_DEVICE(v);
    hipStream_t s = (hipStream_t)stream;
    const bool trusted = done_list_session(v, s);
    if (v->done_list_fresh) v->auto_list = false;  // the last list was never consumed: this caller steps without qg_vec_reset_done
    StepArgs a;
    fill_step_args(v, a);
    const bool lists = trusted && T == 1 && !fused && step_leaves_done_list(v, a);
    if (int rc = lists ? done_list_before_append(v, s) : drop_done_list(v, s)) return rc;
    a.actions = actions_dev;
    a.coins = coins_dev;
    a.rewards_seq = rewards_dev;
    a.dones_seq = dones_dev;
    if (action_dtype == QG_ACT_I64) a.flags |= F_ACT64;
    if (v->layout == LAYOUT_LFD) fused = 0;  // its step kernel spreads an env over four lanes; T steps = T launches (one graph)
    if (dense_rides_in_step(v) && (!fused || T == 1)) a.dense = v->dense;  // single-step launches keep the tracked observation current themselves
    if (fused) {
        if (period != T) return set_error(QG_ERR_INVALID, "fused rollouts read actions[t] for every t");
        a.T = (uint32_t)T;
        HIP_TRY(launch_step(v, a, s));
        v->step_index += T;
        if (v->dense && !a.dense) return dense_refresh(v, s);
        return QG_OK;
    }
    const size_t act_bytes = action_dtype == QG_ACT_I64 ? 8 : 4;
    auto enqueue_steps = [&](hipStream_t st) -> hipError_t {
        for (size_t t = 0; t < T; ++t) {
            StepArgs b = a;
            b.T = 1;
            b.actions = (const char *)actions_dev + (t % period) * v->B * act_bytes;
            b.coins = coins_dev ? coins_dev + t * v->B : nullptr;
            b.rewards_seq = rewards_dev ? rewards_dev + t * v->B : nullptr;
            b.dones_seq = dones_dev ? dones_dev + t * v->B : nullptr;
            b.step_index = v->step_index + t;
            hipError_t e = launch_step(v, b, st);
            if (e != hipSuccess) return e;
        }
        if (v->dense && !a.dense) {  // kernels that do not track it (add_inverts, other layouts): one full rewrite after the last step
            ObsArgs oa;
            fill_obs_args(v, oa, v->dense, QG_FMT_U8, (size_t)v->D * v->D);
            return launch_export(v, oa, st);
        }
        return hipSuccess;
    };
    // A graph bakes kernel arguments in: the counter-RNG coin path changes per call, and a stream
    // that is already being captured by the caller must not be captured again.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s) (void)hipStreamIsCapturing(s, &cs);
    const bool rng_coins = (v->flags & F_INVERTS) && !coins_dev;
    if (cs != hipStreamCaptureStatusNone || rng_coins || T == 1) {
        HIP_TRY(enqueue_steps(s));
        v->step_index += T;
        if (lists) {
            done_list_appended(v, true);
            step_wrote_mask(v, a, true);
        }
        return QG_OK;
    }
    GraphKey key{actions_dev, coins_dev, rewards_dev, dones_dev, T, action_dtype, period, a.flags, v->env_base, v->dense};
    CachedGraph *cg = nullptr;
    for (auto &g : v->graphs)
        if (g.key == key) cg = &g;
    if (!cg) {
        if (!v->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&v->capture_stream, hipStreamNonBlocking));
        CachedGraph ng;
        ng.key = key;
        HIP_TRY(hipStreamBeginCapture(v->capture_stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = enqueue_steps(v->capture_stream);
        hipError_t e2 = hipStreamEndCapture(v->capture_stream, &ng.graph);
        if (e != hipSuccess || e2 != hipSuccess) {
            (void)hipGetLastError();
            if (ng.graph) (void)hipGraphDestroy(ng.graph);
            return set_error(QG_ERR_DEVICE, "graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        }
        HIP_TRY(hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0));
        if (v->graphs.size() >= 8) {  // small LRU-less cache: drop the oldest
            (void)hipGraphExecDestroy(v->graphs.front().exec);
            (void)hipGraphDestroy(v->graphs.front().graph);
            v->graphs.erase(v->graphs.begin());
        }
        v->graphs.push_back(ng);
        cg = &v->graphs.back();
    }
    HIP_TRY(hipGraphLaunch(cg->exec, s));
    v->step_index += T;
    return QG_OK;
}

static int observe_dense_impl(qg_vec *v, int8_t *out_dev, const int32_t *perm_idx_dev, void *stream) {
    QG_ON_DEVICE(v);
    ObsArgs oa;
    qg_vec_info info;
    qg_vec_get_info(v, &info);
    if ((v->layout == LAYOUT_TILE64 || v->layout == LAYOUT_LFD) && info.packed_words_per_env == info.obs_rows && !((uintptr_t)out_dev & 15u)) {
        // 64-bit-row and lane-group layouts: the packed rows (one word per observation row) first, then the shared expansion kernel, whose
        // stores are 16 bytes wide -- the export kernels' byte-at-a-time rows reach 0.9 TB/s (CliffordEnv 24q x 65 536: 168 us for 151 MB)
        const uint64_t n_words = v->B * (uint64_t)info.packed_words_per_env;
        if (int rc = ensure_scratch(v, n_words * info.packed_word_bytes)) return rc;
        fill_obs_args(v, oa, v->scratch, QG_FMT_PACKED, format_min_elems(v, QG_FMT_PACKED));
        HIP_TRY(launch_export(v, oa, (hipStream_t)stream));
        HIP_TRY(expand_rows(v->scratch, (int)info.packed_word_bytes, n_words, (uint32_t)info.obs_cols, out_dev, QG_DT_I8, (hipStream_t)stream));
        v->observe_counter += 1;
        return QG_OK;
    }
    fill_obs_args(v, oa, out_dev, QG_FMT_U8, (size_t)info.obs_rows * info.obs_cols);
    v->perm_draw = true;  // PauliEnv::observe draws a new qubit permutation (pauli.rs:657-662)
    v->perm_in = perm_idx_dev;
    hipError_t e = launch_export(v, oa, (hipStream_t)stream);
    v->perm_draw = false;
    v->perm_in = nullptr;
    v->observe_counter += 1;
    HIP_TRY(e);
    return QG_OK;
}

int qg_vec_observe_dense(qg_vec *v, int8_t *out_dev, void *stream) {
    if (!v || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    return observe_dense_impl(v, out_dev, nullptr, stream);
}

static int dense_refresh(qg_vec *v, hipStream_t s) {
    ObsArgs oa;
    fill_obs_args(v, oa, v->dense, QG_FMT_U8, (size_t)v->D * v->D);
    HIP_TRY(launch_export(v, oa, s));
    return QG_OK;
}

int qg_vec_track_dense(qg_vec *v, int8_t *dense_dev, void *stream) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    if (!dense_dev) {
        v->dense = nullptr;
        return QG_OK;
    }
    if (!plan::dense_trackable(plan_of(v)))
        return set_error(QG_ERR_UNSUPPORTED, "track_dense: matrices of 16 or 32 rows held as 32-bit row words (CliffordEnv N = 8, 16; LinearFunctionEnv N = 16, 32)");
    if ((uintptr_t)dense_dev & 15u) return set_error(QG_ERR_INVALID, "track_dense: the buffer must be 16-byte aligned");
    v->dense = dense_dev;
    return dense_refresh(v, (hipStream_t)stream);
}

int qg_plan_query(const qg_config *cfg, uint64_t batch, uint32_t num_actions, int op, uint64_t arg, int nonsymplectic, char *name_out, size_t cap) {
    if (!cfg || !name_out || !cap) return set_error(QG_ERR_INVALID, "null argument");
    if (cfg->num_qubits <= 0 || batch == 0) return set_error(QG_ERR_INVALID, "num_qubits and batch must be positive");
    plan::HandlePlan hp;
    const char *why = "";
    if (int rc = plan::handle_plan(*cfg, batch, hp, why)) return set_error(rc, "%s", why);
    const char *name = "(none)";
    static const char *layouts[] = {"NONE", "?", "LF8", "PERM", "PAULI", "TILE", "TILE64", "PERMB", "LFD"};
    const bool perms = cfg->env_kind == QG_PAULI && cfg->add_perms;  // (a PauliEnv whose coupling map has automorphisms)
    const uint32_t R = hp.has_z ? 2 * hp.nxp : hp.nxp;
    switch (op) {
    case QG_PLAN_LAYOUT: name = hp.layout == LAYOUT_PAULI ? (hp.pauli_compact ? "PTILE-compact" : "PTILE") : layouts[hp.layout]; break;
    case QG_PLAN_STEP: name = plan::step_kernel_name(plan::step_kernel_of(hp, 1, false, false, nonsymplectic != 0, num_actions, perms)); break;
    case QG_PLAN_ROLLOUT_FUSED:
        name = plan::step_kernel_name(plan::step_kernel_of(hp, (uint32_t)(arg ? arg : 1), true, true, nonsymplectic != 0, num_actions, perms));
        break;
    case QG_PLAN_RESET_DONE: {
        const uint32_t draws = (uint32_t)cfg->difficulty, count = (uint32_t)arg;
        if (hp.layout == LAYOUT_TILE) {
            const bool coop = plan::reset_coop_allowed(false, batch, true);
            name = plan::reset_path_name(plan::list_reset_path(count, draws, batch, coop, plan::tile_coop_fits(R, 4)));
        } else if (hp.layout == LAYOUT_TILE64) {
            const bool coop = plan::reset_coop_allowed(false, batch, true);
            const plan::ResetPath rp = plan::list_reset_path(count, draws, batch, coop, true);
            name = rp == plan::RP_TREE ? "scramble_tree64" : plan::reset_path_name(rp);
        } else if (hp.layout == LAYOUT_PAULI) {
            // (the gateset's CX count is not known here: a PauliEnv gateset is taken to hold at least one CX and at most num_actions of them)
            name = batch > QG_COMPACT_MIN_ENVS ? (plan::pauli_tree_takes(count, draws, batch, (uint32_t)num_actions) ? "compact_done + ptile_reset_tree_kernel"
                                                                                                                       : "compact_done + ptile_generate_kernel")
                                               : "ptile_generate_kernel";
        } else if (hp.layout == LAYOUT_LF8 || hp.layout == LAYOUT_PERM) {
            name = "word_init_kernel";  // (decides per wave: 16 lanes per finished env up to 8 of them, the per-lane chain in a fuller wave)
        } else {
            name = "init_kernel";
        }
        break;
    }
    case QG_PLAN_RESET_DONE_STEP:
        name = (plan::reset_step_fusable(hp) && plan::reset_step_pays(hp, cfg->difficulty, std::min<int64_t>((int64_t)cfg->depth_slope * cfg->difficulty, cfg->max_depth)))
                   ? ((hp.flags & F_INVERTS) ? (nonsymplectic ? "two launches" : "qm_reset_inv2_step_kernel (after a list-leaving step)")
                                               : hp.layout == LAYOUT_TILE64 ? "q64_reset_step_kernel (after a list-leaving step)" : "qm_reset_step_kernel (after a list-leaving step)")
               : plan::reset_step_in_word_kernel(hp, num_actions) ? "word_reset_step_kernel" : "two launches";
        break;
    case QG_PLAN_OBSERVE_DENSE:
        if (hp.layout == LAYOUT_TILE) name = plan::export_kernel_name(plan::tile_export(QG_FMT_U8, hp.D, R, (uint64_t)hp.D * hp.D, true, true));
        else if (hp.layout == LAYOUT_TILE64 || hp.layout == LAYOUT_LFD || hp.layout == LAYOUT_PAULI) name = plan::export_kernel_name(plan::EK_WORDS_THEN_EXPAND);
        else name = plan::export_kernel_name(plan::EK_GENERIC);
        break;
    case QG_PLAN_OBSERVE_PACKED:
        name = hp.layout == LAYOUT_TILE ? plan::export_kernel_name(plan::tile_export(QG_FMT_PACKED, hp.D, R, hp.D, true, true))
               : hp.layout == LAYOUT_PAULI ? "ptile_rowwords_kernel" : plan::export_kernel_name(plan::EK_GENERIC);
        break;
    case QG_PLAN_STATE_I64:
        name = (plan::entry_formats_stream(hp.layout, batch) || (hp.layout == LAYOUT_PAULI && batch >= QG_STREAM_MIN_ENVS)) ? "row words / bit stream + streaming kernel"
                                                                                                                  : "init / export kernel";
        break;
    case QG_PLAN_TRACK_DENSE:
        if (!plan::dense_trackable(hp)) return set_error(QG_ERR_UNSUPPORTED, "track_dense: matrices of 16 or 32 rows held as 32-bit row words");
        name = plan::dense_rides_in_step(hp) ? "in-step" : "refresh";
        break;
    default: return set_error(QG_ERR_INVALID, "unknown plan op %d", op);
    }
    snprintf(name_out, cap, "%s", name);
    return QG_OK;
}

int qg_vec_pauli_observe_dense(qg_vec *v, int8_t *out_dev, const int32_t *perm_idx_dev, void *stream) {
    if (!v || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (v->layout != LAYOUT_PAULI) return set_error(QG_ERR_INVALID, "not a PauliEnv batch");
    return observe_dense_impl(v, out_dev, perm_idx_dev, stream);
}

int qg_vec_pauli_num_perms(const qg_vec *v) { return v ? (int)v->n_perms : -1; }

int qg_vec_observe_packed(qg_vec *v, void *out_dev, void *stream) {
    if (!v || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    if (v->layout == LAYOUT_PAULI) {
        const uint32_t cols = 2 * v->N + (uint32_t)std::max(v->cfg.max_rotations, 1);
        if (cols > 64u)
            return set_error(QG_ERR_UNSUPPORTED, "packed observation of PauliEnv needs at most 64 observation columns: use observe_dense");
        v->perm_draw = true;  // PauliEnv::observe draws a new qubit permutation (pauli.rs:657-662)
        const hipError_t e = ptile_observe_words(v, out_dev, (hipStream_t)stream);
        v->perm_draw = false;
        v->observe_counter += 1;
        HIP_TRY(e);
        return QG_OK;
    }
    ObsArgs oa;
    fill_obs_args(v, oa, out_dev, QG_FMT_PACKED, format_min_elems(v, QG_FMT_PACKED));
    HIP_TRY(launch_export(v, oa, (hipStream_t)stream));
    return QG_OK;
}

static int observe_host(qg_vec *v, void *out_host, bool packed, void *stream) {
    if (!v || !out_host) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    qg_vec_info info;
    qg_vec_get_info(v, &info);
    const size_t bytes = packed ? (size_t)v->B * info.packed_words_per_env * info.packed_word_bytes : (size_t)v->B * info.obs_rows * info.obs_cols;
    if (v->host_obs_bytes < bytes) {
        if (v->host_obs) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(v->host_obs));
            v->host_obs = nullptr;
            v->host_obs_bytes = 0;
        }
        HIP_TRY(hipMalloc(&v->host_obs, bytes));
        v->host_obs_bytes = bytes;
    }
    if (int rc = packed ? qg_vec_observe_packed(v, v->host_obs, stream) : qg_vec_observe_dense(v, (int8_t *)v->host_obs, stream)) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, v->host_obs, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return QG_OK;
}
int qg_vec_observe_dense_host(qg_vec *v, int8_t *out_host, void *stream) { return observe_host(v, out_host, false, stream); }
int qg_vec_observe_packed_host(qg_vec *v, void *out_host, void *stream) { return observe_host(v, out_host, true, stream); }

int qg_vec_masks(qg_vec *v, uint8_t *out_dev, void *stream) {
    if (!v || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    HIP_TRY(masks_fill(v->success, out_dev, v->B, (uint32_t)v->gates.size(), (hipStream_t)stream));
    return QG_OK;
}

int qg_vec_pauli_reset_from(qg_vec *v, const uint8_t *tableaus, const char *labels, const int32_t *n_rot, void *stream) {
    if (!v || !tableaus || !n_rot) return set_error(QG_ERR_INVALID, "null argument");
    if (v->layout != LAYOUT_PAULI) return set_error(QG_ERR_INVALID, "not a PauliEnv batch");
    QG_ON_DEVICE(v);
    return pauli_reset_from(v, tableaus, labels, n_rot, (hipStream_t)stream);
}

int qg_vec_sync(qg_vec *v, void *stream) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    if (!v->fault_word) HIP_TRY(hipHostMalloc((void **)&v->fault_word, sizeof(uint32_t), hipHostMallocMapped));
    if (!v->fault_scratch) {
        HIP_TRY(hipMalloc((void **)&v->fault_scratch, 2 * sizeof(uint32_t)));
        HIP_TRY(hipMemset(v->fault_scratch, 0, 2 * sizeof(uint32_t)));
    }
    *v->fault_word = 0xFFFFFFFFu;  // whatever goes wrong below reads as "look at the array"
    HIP_TRY(fault_any(v->error, v->B, v->fault_scratch, v->fault_word, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (*(volatile uint32_t *)v->fault_word == 0) return QG_OK;
    std::vector<uint32_t> err(v->B);
    HIP_TRY(hipMemcpy(err.data(), v->error, sizeof(uint32_t) * v->B, hipMemcpyDeviceToHost));
    for (uint64_t e = 0; e < v->B; ++e)
        if (err[e]) {
            const char *what = (err[e] & QG_FAULT_SINGULAR)      ? "singular matrix in inverse() (reference panics, clifford.rs:155)"
                               : (err[e] & QG_FAULT_ZERO_WEIGHT) ? "weight-0 rotation in the front layer (reference panics, pauli_network.rs:114)"
                               : (err[e] & QG_FAULT_BAD_STATE)   ? "set_state produced an unusable state"
                                                                 : "solution log overflow (QG_FAULT_SOLUTION_OVERFLOW)";
            return set_error(QG_ERR_PANIC, "env %llu: %s (fault bits 0x%x)", (unsigned long long)e, what, err[e]);
        }
    return QG_OK;
}

// Env::solution from one env's slice of the log (`row[i * stride]` = entry i in push order, `len` = {pushed to solution, to solution_inv})
static size_t decode_solution(const qg_vec *v, const uint32_t *row, size_t stride, const int32_t len[2], uint64_t *out, size_t cap) {
    size_t n = 0;
    if (v->layout == LAYOUT_PAULI) {  // one list (pauli.rs:685-719); 32-bit entries
        for (int32_t i = 0; i < len[0]; ++i, ++n)
            if (n < cap && out) out[n] = row[i * stride] == 0xFFFFFFFFu ? ~0ull : (uint64_t)row[i * stride];  // saturated invalid action
        return n;
    }
    // solution ++ reverse(solution_inv) (clifford.rs:376-381).  The log holds the pushes in order, bit 31 = pushed to solution_inv.
    auto widen = [](uint32_t w) { return (w & 0x7FFFFFFFu) == 0x7FFFFFFFu ? ~0ull : (uint64_t)(w & 0x7FFFFFFFu); };
    const uint32_t total = (uint32_t)std::min<int64_t>((int64_t)len[0] + len[1], v->sol_cap);
    for (uint32_t i = 0; i < total; ++i)
        if (!(row[i * stride] >> 31)) {
            if (n < cap && out) out[n] = widen(row[i * stride]);
            ++n;
        }
    for (uint32_t i = total; i-- > 0;)
        if (row[i * stride] >> 31) {
            if (n < cap && out) out[n] = widen(row[i * stride]);
            ++n;
        }
    return n;
}

int64_t qg_vec_solution(qg_vec *v, uint64_t env, uint64_t *out, size_t cap) {
    if (!v) return set_error(QG_ERR_INVALID, "null argument");
    if (env >= v->B) return set_error(QG_ERR_INVALID, "env index out of range");
    if (!v->cfg.track_solution) return 0;
    QG_ON_DEVICE(v);
    int32_t len[2];
    std::vector<uint32_t> row(v->sol_cap);
    // the log is step-major (entry `slot` of env e at sol[slot * B + e]): one env's entries are a strided column
    if (hipMemcpy(len, v->sol_len + env * 2, sizeof len, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy2D(row.data(), sizeof(uint32_t), v->sol + env, sizeof(uint32_t) * v->B, sizeof(uint32_t), v->sol_cap, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "solution copy failed");
    }
    return (int64_t)decode_solution(v, row.data(), 1, len, out, cap);
}

int qg_vec_solutions(qg_vec *v, uint64_t *out, size_t cap, int64_t *lens) {
    if (!v || !lens || (cap && !out)) return set_error(QG_ERR_INVALID, "null argument");
    if (!v->cfg.track_solution) {
        std::fill(lens, lens + v->B, (int64_t)0);
        return QG_OK;
    }
    QG_ON_DEVICE(v);
    std::vector<int32_t> len((size_t)v->B * 2);
    std::vector<uint32_t> log((size_t)v->sol_cap * v->B);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(len.data(), v->sol_len, len.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(log.data(), v->sol, log.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "solution copy failed");
    }
    for (size_t e = 0; e < v->B; ++e) lens[e] = (int64_t)decode_solution(v, log.data() + e, v->B, &len[2 * e], out ? out + e * cap : nullptr, cap);
    return QG_OK;
}

}  // extern "C"

namespace qg {
void fill_step_args_public(const qg_vec *v, StepArgs &a) { fill_step_args(v, a); }
int bind_error(qg_vec *v, uint32_t *error_dev) {
    if (!v || !error_dev) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(v);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(error_dev, v->error, sizeof(uint32_t) * v->B, hipMemcpyDefault));
    if (v->own_error) (void)hipFree(v->error);
    v->error = error_dev;
    v->own_error = false;
    drop_graphs(v);
    return QG_OK;
}
unsigned long long *kernel_clock_slot_public(const qg_vec *v) { return kernel_clock_slot(v); }
uint32_t reset_tree_grid_public(const qg_vec *v, uint32_t most) { return reset_tree_grid(v, most); }
// The launch behind a reset's trees (PauliEnv's ptile_generate_kernel) has nothing to do when the trees took the list, and
// finds that out from one word: a few workgroups are enough for it whenever the latest list the handle's resets have reported was a trees' list (they walk the batch
// with the grid's stride, so a longer list is still reset -- by fewer workgroups, until the next call sees its length).  0: one workgroup per 64 envs.
uint32_t reset_second_grid_public(const qg_vec *v, bool is_tree_list_of_that_length(uint32_t, const qg_vec *)) {
    const uint32_t seen = v->count_seen ? *(volatile const uint32_t *)v->count_seen : 0xFFFFFFFFu;
    return (seen != 0xFFFFFFFFu && (seen == 0 || is_tree_list_of_that_length(seen, v))) ? 32u : 0u;
}

int dense_refresh_public(qg_vec *v, hipStream_t s) { return dense_refresh(v, s); }
// InitArgs of qg_vec_reset_done(v, seed) without a list: what a kernel that resets finished envs itself needs (qg_vec_mid_head_sample_step)
void fill_reset_done_args_public(const qg_vec *v, uint64_t seed, InitArgs &ia) {
    fill_init_args(v, ia);
    ia.mode = 2;
    ia.n_draws = (uint32_t)v->difficulty;
    ia.seed = seed;
    ia.only_done = 1u;
    const int64_t d = (int64_t)v->cfg.depth_slope * v->difficulty;  // clifford.rs:317
    ia.depth_value = (int32_t)std::min<int64_t>(d, v->cfg.max_depth);
}
int reset_done_public(qg_vec *v, uint64_t seed, void *stream) { return qg_vec_reset_done(v, seed, stream); }
}  // namespace qg
