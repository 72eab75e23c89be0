// qgym_host.hpp -- host-side handle of a batched env (private to libqgym).
#pragma once

#include <string>
#include <vector>

#include "qgym_internal.hpp"
#include "qgym_plan.hpp"

namespace qg {

struct GraphKey {
    const void *actions;
    const void *coins;
    const void *rewards;
    const void *dones;
    size_t T;
    int dtype;
    size_t period;
    uint32_t flags;     // effective StepArgs::flags: they pick the kernel variant (e.g. F_GJ follows maybe_nonsymplectic)
    uint64_t env_base;  // baked into the captured launches like every other kernel argument
    const void *dense;  // qg_vec_track_dense: the tracked buffer (picks the DENSE kernel variant, and is baked in)
    bool operator==(const GraphKey &o) const {
        return actions == o.actions && coins == o.coins && rewards == o.rewards && dones == o.dones && T == o.T &&
               dtype == o.dtype && period == o.period && flags == o.flags && env_base == o.env_base && dense == o.dense;
    }
};
struct CachedGraph {
    GraphKey key{};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

int set_error(int code, const char *fmt, ...);

// HIP call inside a function that returns a qg_status
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (void)hipGetLastError();                                                               \
            return set_error(QG_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e));        \
        }                                                                                          \
    } while (0)

// Every C entry point works on its handle's GPU and leaves the calling thread's current device as it found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define QG_ON_DEVICE(v)                                                                            \
    qg::DeviceGuard _qg_guard((v)->device);                                                        \
    if (_qg_guard.err != hipSuccess) {                                                             \
        (void)hipGetLastError();                                                                   \
        return qg::set_error(QG_ERR_DEVICE, "cannot select device %d: %s", (v)->device, hipGetErrorString(_qg_guard.err)); \
    }

}  // namespace qg

struct qg_vec {
    qg_config cfg{};
    std::vector<qg_gate> gates;
    uint64_t B = 0;
    uint32_t N = 0, D = 0, log2L = 0;
    int device = 0;
    qg::Layout layout = qg::LAYOUT_NONE;
    size_t stride_bytes = 0;   // per-env stride (0 for the tiled layout)
    size_t state_bytes = 0;    // total resident state size
    uint32_t nxp = 0;          // TILE layout: X-row slots per env (N rounded up to 4); PERMB layout: 16-byte groups per env
    bool has_z = false;        // TILE layout: Z-type rows present (CliffordEnv)
    bool w64 = false;          // LFD layout: uint64 rows (N > 32); nxp = groups per region
    uint32_t flags = 0;
    int64_t difficulty = 1;
    uint64_t coin_seed = 0;
    uint64_t step_index = 0;
    uint64_t env_base = 0;  // qg_vec_set_env_base: global index of env 0 in every counter-RNG draw (a shard of a larger batch)
    const uint64_t *clock_dev = nullptr;  // qg_vec_set_clock (not owned)
    // qg_vec_set_kernel_clock (not owned): [kclk_cap][kclk_waves][2] device words; the k-th step / observation launch after the call stamps slot k
    unsigned long long *kclk = nullptr;
    size_t kclk_cap = 0;
    uint32_t kclk_waves = 0;
    mutable size_t kclk_next = 0;

    // device buffers
    void *state = nullptr;
    int32_t *depth = nullptr;
    float *reward = nullptr;
    uint8_t *done = nullptr;
    uint8_t *success = nullptr;
    uint8_t *inverted = nullptr;
    uint32_t *error = nullptr;
    uint32_t *sol = nullptr;
    int32_t *sol_len = nullptr;
    int32_t *layers = nullptr;
    qg::GateEntry *d_gates = nullptr;
    uint32_t *d_descs = nullptr;
    uint32_t sol_cap = 0;
    uint32_t layers_len = 0;
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *host_in = nullptr;            // qg_vec_step_host: device copies of the caller's actions [B] (8 bytes each) and coins [B]
    void *host_obs = nullptr;           // qg_vec_observe_*_host: the observation before its copy to the caller's buffer
    size_t host_obs_bytes = 0;
    uint32_t *count_seen = nullptr;     // TILE: how many envs the latest finished reset_done launch reset (pinned, device-mapped; written by the launch, read -- without
                                        // waiting -- when a later one is sized: qgym_api.cpp reset_tree_grid).  0xFFFFFFFF: nothing seen yet
    uint32_t *fault_word = nullptr;     // qg_vec_sync: OR of error[], one pinned, device-mapped word the host reads after the stream has drained
    uint32_t *fault_scratch = nullptr;  // its device-side accumulator and ticket (two words, zero between calls)
    uint32_t *bad = nullptr;            // TILE / TILE64 without add_inverts: per-env "differs from identity" mask (one-step kernels)
    uint32_t *d_rowops = nullptr;       // TILE: gate table as pairs of row operations on slots (cooperative reset kernel)
    uint32_t *done_list = nullptr;      // reset_done: [B] indices of finished envs + {length, reader ticket} at [B], [B + 1]
    // qg_vec_reset_done_step (TILE without add_inverts): the fused launch consumes done_list and appends to done_list_alt, then the two trade
    // places
    uint32_t *done_list_alt = nullptr;
    uint32_t *done_list_spare = nullptr;  // TILE: the list no launch in flight reads or appends to; the reset that consumes done_list zeroes this one's length, then they rotate
    // TILE: the finished envs of a step as one bit per env (StepArgs::done_mask), two buffers -- a list-leaving step (or the fused reset + step
    // launch, which reads the current one) writes the other one, then they trade places.  Nothing to zero: a launch rewrites every word.
    uint64_t *done_mask[2] = {nullptr, nullptr};
    uint32_t *mask_count = nullptr;     // TILE64 / PauliEnv: a device word where a reset's first launch leaves the mask's count for its second
    int mask_cur = 0;
    uint32_t mask_epoch[2] = {0, 0};    // StepArgs::done_epoch of the launch that wrote each buffer (InitArgs::mask_epoch for its reader)
    bool mask_fresh = false;            // done_mask[mask_cur] (+ the list in done_list: envs reset and final again inside the fused launch; else empty) holds
                                        // the envs that are final, as the handle's own last step left them (believed within the session, like done_list_fresh)
    bool alt_zero_known = true;         // done_list_alt's length is known to be zero
    bool auto_list = false;             // qg_vec_reset_done is in use on this handle: single steps append the envs they finish to the list themselves
    bool done_list_fresh = false;       // the list already holds the finished envs (written by the step that ended them); believed within the session only
    bool list_zero_known = true;        // the list's length is known to be zero (creation, a memset, or its consumer ran) -- within the session
    bool list_tainted = false;          // some launch that touches the list was captured into a caller's graph: eager calls trust nothing
    uint64_t list_session = 0;          // 0 = eager execution, else the stream capture id the beliefs above belong to (qgym_api.cpp)
    int8_t *dense = nullptr;            // qg_vec_track_dense: caller-owned [B][rows][cols] int8 observation kept equal to observe_dense() of the state
    uint32_t *d_nonsymp = nullptr;      // device word behind InitArgs::nonsymp_flag
    void *embed_dump = nullptr;         // qg_vec_embed: 1 KiB nobody reads (kernels_policy.hip), allocated by qg_vec_pack_embedding
    bool maybe_nonsymplectic = false;   // CliffordEnv + add_inverts: some env may need the Gauss-Jordan inversion
    bool own_reward = true, own_done = true, own_success = true, own_depth = true;
    bool own_error = true;  // (the scalar env keeps its one fault word in its pinned I/O block: qg::bind_error)

    // PauliEnv (pauli_host.cpp, kernels_pauli_tile.hip)
    void *d_prog = nullptr;  // [num_actions] per-action programs (tableau map + micro-ops)
    uint32_t rmax = 0;
    uint32_t rmax_generate = 0;
    uint32_t pt_nq = 0, pt_rm = 0;
    void *d_gen_tables = nullptr;  // PTILE target generator: coupling-graph distance tables
    uint32_t gen_nd = 0, gen_ncx = 0, gen_npairs = 0, gen_off[4] = {0, 0, 0, 0};  // final_pauli_layers: most rotations reset() generates
    uint8_t *d_qubit_perms = nullptr;  // [n_perms][N]  (add_perms)
    int32_t *d_act_perms = nullptr;    // [n_perms][num_actions]
    uint32_t *perm_idx = nullptr;      // [B] current_perm_idx
    const int32_t *perm_in = nullptr;  // explicit draws for the next observe (not owned)
    uint32_t n_perms = 0;
    bool perm_draw = false;            // the export in flight is an observe() (draws a new perm)
    uint64_t observe_counter = 0;

    // rollout graphs
    std::vector<qg::CachedGraph> graphs;
    hipStream_t capture_stream = nullptr;
};

namespace qg {
int ensure_scratch_public(qg_vec *v, size_t bytes);
// done-list protocol (qgym_api.cpp): session scoping of the host's beliefs, zeroing before an appending launch
bool done_list_session(qg_vec *v, hipStream_t s);
int done_list_before_append(qg_vec *v, hipStream_t s);
void done_list_appended(qg_vec *v, bool trusted);
void fill_step_args_public(const qg_vec *v, StepArgs &a);
// the per-env fault words in caller-owned memory (device-visible, [B] uint32, current content carried over); not part of the C ABI: the scalar env's
int bind_error(qg_vec *v, uint32_t *error_dev);
uint32_t reset_second_grid_public(const qg_vec *v, bool is_tree_list_of_that_length(uint32_t, const qg_vec *));  // workgroups of the launch behind a reset's trees
uint32_t reset_tree_grid_public(const qg_vec *v, uint32_t most);  // workgroups of a reset's tree launch, from the list lengths the handle's resets have reported
unsigned long long *kernel_clock_slot_public(const qg_vec *v);  // qg_vec_set_kernel_clock: the slot of the launch about to be enqueued, or null
// qg_vec_track_dense: rewrite the whole tracked observation from the state (after a launch that changed states without updating it)
int dense_refresh_public(qg_vec *v, hipStream_t s);
void fill_reset_done_args_public(const qg_vec *v, uint64_t seed, InitArgs &ia);
void compute_qubit_and_action_perms(uint32_t N, const std::vector<qg_gate> &gates, std::vector<std::vector<int64_t>> &qubit_perms,
                                    std::vector<std::vector<int64_t>> &act_perms);
// PauliEnv host hooks (pauli_host.cpp)
int pauli_alloc(qg_vec *v);
int pauli_init_identity(qg_vec *v, hipStream_t s);
int pauli_set_state(qg_vec *v, const void *states, int format, size_t stride, int on_device, hipStream_t s);
int pauli_reset_from(qg_vec *v, const uint8_t *tableaus, const char *labels, const int32_t *n_rot, hipStream_t s);
int pauli_reset_seeded(qg_vec *v, uint64_t seed, hipStream_t s);
hipError_t pauli_step(const qg_vec *v, const StepArgs &a, hipStream_t s);
hipError_t pauli_export(const qg_vec *v, const ObsArgs &a, hipStream_t s);
}  // namespace qg
