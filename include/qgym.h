/*
 * qgym.h -- C ABI of the MI355X-native batched env.step() path (libqgym.so).
 *
 * This is the drop-in boundary for qiskit-gym's Rust `env.step()` hot path.  The reference
 * exposes each environment to its callers through the `twisterl::rl::env::Env` trait
 * (implemented at rust/src/envs/clifford.rs:285-382, linear_function.rs:259-365,
 * permutation.rs:148-257, pauli.rs:492-720 of the reference) and to Python through PyO3 classes
 * registered in rust/src/lib.rs:24-31.  A Rust (or cgo / ctypes) host binds the functions below
 * instead; INTEGRATION.md shows the `impl Env` shim a maintainer would add.
 *
 * Two flavours:
 *   qg_vec_*  a batch of B independent environments resident in one GPU's HBM, stepped by one
 *             kernel launch per `step` (new surface: the reference has no vector env, every
 *             method is the batched counterpart of one trait method and cites it);
 *   qg_env_*  one environment with exactly the trait's method set (a batch of 1 on the same
 *             kernels), for callers that hold a `Box<dyn Env>`.
 *
 * Conventions: plain pointers and sizes only.  Pointers named *_dev are device pointers on the
 * handle's GPU; `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream).  All
 * qg_vec_* device work is enqueued on `stream` and returns without synchronising unless stated.
 * Every function returns QG_OK (0) or a negative qg_status; qg_last_error() gives the message
 * (thread-local).  There is no CPU fallback: without a visible gfx950 device creation fails.
 */
#ifndef QGYM_H
#define QGYM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: + qg_comm_* / learner-shard entry points, qg_vec_step_host, qg_vec_observe_*_host (additions only: version-1 callers keep working)
 * 3: + qg_vec_track_dense, qg_comm_p2p_reset, qg_plan_query, qg_vec_reset_done_step, qg_env_pool_clear (additions only) */
#define QG_ABI_VERSION 3

typedef enum {
    QG_OK = 0,
    QG_ERR_INVALID = -1,     /* bad argument (PyValueError in the reference's ctor path) */
    QG_ERR_TYPE = -2,        /* malformed gate (PyTypeError, common.rs:51-76) */
    QG_ERR_UNSUPPORTED = -3, /* outside the limits documented in DESIGN.md */
    QG_ERR_DEVICE = -4,      /* HIP runtime failure / no GPU */
    QG_ERR_PANIC = -5        /* the reference would have panicked (singular inverse, malformed Pauli state, ...) */
} qg_status;

/* rust/src/envs/common.rs:19-29 (enum order) */
typedef enum { QG_H = 0, QG_S = 1, QG_SDG = 2, QG_SX = 3, QG_SXDG = 4, QG_CX = 5, QG_CZ = 6, QG_SWAP = 7 } qg_gate_kind;

/* the four `impl Env` blocks; numbering shared with the oracle */
typedef enum { QG_PERMUTATION = 0, QG_LINEAR_FUNCTION = 1, QG_CLIFFORD = 2, QG_PAULI = 3 } qg_env_kind;

typedef struct {
    int32_t kind; /* qg_gate_kind */
    int32_t q0;
    int32_t q1; /* ignored for one-qubit gates */
} qg_gate;

/* Constructor arguments of Py{Clifford,LinearFunction,Permutation,Pauli}Env::new
 * (clifford.rs:401-426, linear_function.rs:384-410, permutation.rs:277-303, pauli.rs:743-778). */
typedef struct {
    int32_t env_kind; /* qg_env_kind */
    int32_t num_qubits;
    int32_t difficulty;
    int32_t depth_slope;
    int32_t max_depth;
    /* MetricsWeights (metrics.rs:150-166) */
    float w_n_cnots, w_n_layers_cnots, w_n_layers, w_n_gates;
    int32_t add_inverts;
    int32_t add_perms;
    int32_t track_solution;
    /* PauliEnv only */
    int32_t max_rotations;
    int32_t pauli_diff_scale;
    int32_t final_pauli_layers; /* < 0: None -> max_rotations + 2 (pauli.rs:760) */
    float num_qubits_decay;
    float pauli_layer_reward;
} qg_config;

/* Fill the reference's defaults for `env_kind` (clifford.rs:420-422, metrics.rs:157-166,
 * envs/synthesis.py:182-204,380-412). */
void qg_config_default(qg_config *cfg, int32_t env_kind, int32_t num_qubits);

const char *qg_last_error(void);
int qg_abi_version(void);
/* Number of visible gfx950 devices (0 when none / no driver). Never fails. */
int qg_device_count(void);

/* Parse one gate given as (name, indices) -- the FromPyObject of common.rs:46-100: name is
 * trimmed and matched case-insensitively; wrong arity / unknown name -> QG_ERR_INVALID with the
 * reference's message. */
int qg_gate_parse(const char *name, const int64_t *indices, size_t n_indices, qg_gate *out);

/* ------------------------------------------------------------------------------------------
 * Batched environments
 * ---------------------------------------------------------------------------------------- */
typedef struct qg_vec qg_vec;

/* Formats understood by qg_vec_set_state / qg_vec_get_state. */
typedef enum {
    QG_FMT_I64 = 0,   /* the trait's `set_state(Vec<i64>)` wire format per env (clifford.rs:299-304:
                         D*D entries, >0 => 1; permutation.rs:168-173: N entries; pauli.rs:517-552:
                         [rot_count, 4N^2 tableau, (len, chars...)*], fixed stride per env) */
    QG_FMT_U8 = 1,    /* dense 0/1 bytes in observation layout (the Gym int8 observation) */
    QG_FMT_PACKED = 2 /* bit-packed rows, reference row order: row r of env e is word e*D + r;
                         word = uint32 when D <= 32 else uint64; bit c = entry (r, c).
                         Permutation: one uint8 per entry. */
} qg_state_format;

/* Action tensor element type for qg_vec_step / qg_vec_rollout. */
typedef enum { QG_ACT_I32 = 0, QG_ACT_I64 = 1 } qg_action_dtype;

typedef struct {
    int32_t env_kind;
    int32_t num_qubits;
    int32_t num_actions;   /* Env::num_actions (clifford.rs:289) */
    int32_t obs_rows;      /* Env::obs_shape (clifford.rs:291-294, pauli.rs:505-507) */
    int32_t obs_cols;
    int32_t device;
    uint64_t batch;
    uint32_t packed_word_bytes;   /* 4 or 8 (1 for Permutation) */
    uint32_t packed_words_per_env; /* D (N for Permutation) */
    uint64_t packed_env_stride_bytes; /* stride of one env in the resident packed state */
    /* resident device buffers (owned by the handle; valid until destroy) */
    void *state_dev;    /* resident packed state, layout in DESIGN.md section 3 */
    float *reward_dev;  /* [B] Env::reward   (clifford.rs:355) */
    uint8_t *done_dev;  /* [B] Env::is_final (clifford.rs:353) */
    uint8_t *success_dev; /* [B] Env::success (clifford.rs:357-359) */
    int32_t *depth_dev; /* [B] remaining depth */
    uint32_t *error_dev; /* [B] sticky per-env fault bits (QG_FAULT_*) */
} qg_vec_info;

#define QG_FAULT_SINGULAR 1u     /* inverse of a singular matrix requested (clifford.rs:155 panic) */
#define QG_FAULT_ZERO_WEIGHT 2u  /* weight-0 rotation in the front layer (pauli_network.rs:114 unwrap) */
#define QG_FAULT_BAD_STATE 4u    /* set_state produced an unusable state: an out-of-range PermutationEnv entry (the reference indexes out
                                    of bounds, permutation.rs:241-243), or -- with add_inverts only -- a repeated entry, which has no inverse;
                                    without add_inverts a vector with repeated entries runs as in the reference (permutation.rs:168-173) */
#define QG_FAULT_SOLUTION_OVERFLOW 8u /* track_solution: more entries than an episode can produce were logged (the
                                    log holds the max_depth steps an episode can last -- PauliEnv: plus one entry per
                                    rotation; the reference's Vec grows without bound when a caller keeps stepping a
                                    finished env) */

/* Build B environments, all in the constructor state (identity, depth 1, success, reward 1.0;
 * clifford.rs:214-245), on GPU `device`. */
int qg_vec_create(const qg_config *cfg, const qg_gate *gates, size_t n_gates, uint64_t batch, int device, qg_vec **out);
void qg_vec_destroy(qg_vec *v);
int qg_vec_get_info(const qg_vec *v, qg_vec_info *out);

/* Make the handle keep its per-env result arrays in caller-owned device memory (e.g. tensors of
 * the learner's framework) instead of its own: reward f32[B], done u8[B], success u8[B],
 * depth i32[B].  Current contents are carried over.  The caller keeps the memory alive until the
 * handle is destroyed or re-bound; NULL leaves that array where it is.  Synchronises. */
int qg_vec_bind_outputs(qg_vec *v, float *reward_dev, uint8_t *done_dev, uint8_t *success_dev, int32_t *depth_dev);

/* Env::set_difficulty / get_difficulty (clifford.rs:296-297) -- one value for the whole batch */
int qg_vec_set_difficulty(qg_vec *v, int64_t difficulty);
int64_t qg_vec_get_difficulty(const qg_vec *v);

/* Env::set_state for every env (clifford.rs:299-304): states + e*stride_elems is env e's record in
 * `format`; depth := max_depth, metrics zeroed, reward := success ? 1 : 0, inverted := false.
 * `on_device` != 0: `states` is a device pointer.  Host input is staged and copied on `stream`. */
int qg_vec_set_state(qg_vec *v, const void *states, int format, size_t stride_elems, int on_device, void *stream);
/* Inverse of the above for inspection / checkpointing (QG_FMT_I64 of a PauliEnv returns the
 * tableau only). */
int qg_vec_get_state(qg_vec *v, void *out, int format, size_t stride_elems, int on_device, void *stream);

/* Env::reset for every env (clifford.rs:306-319).  The reference draws `difficulty` uniform
 * actions from an unseedable RNG; here draw t of env e is
 *   action = mulhi64(splitmix64(seed ^ splitmix64(e * 0x9E3779B97F4A7C15 + t)), num_actions)
 * so the oracle can replay it.  (PauliEnv: see qg_vec_pauli_reset_from.) */
int qg_vec_reset(qg_vec *v, uint64_t seed, void *stream);
/* Env::reset for the envs whose episode is over (done flag set) only; the others are untouched.
 * Lets a GPU-resident collector run episode after episode without a host round trip; pass a
 * fresh seed per call (e.g. a step counter) so successive episodes of an env differ. */
int qg_vec_reset_done(qg_vec *v, uint64_t seed, void *stream);
/* qg_vec_reset_done(v, reset_seed) followed by qg_vec_step(v, actions_dev, ...) -- the auto-reset collection loop's pair of calls -- with the
 * results of exactly those two calls (rewards_dev / dones_dev: optional per-step outputs as in qg_vec_rollout).  ONE launch from the second
 * call of a session on for: CliffordEnv N <= 16 with any options (add_inverts included, while every env is symplectic and no dense observation
 * is tracked with it), CliffordEnv 16 < N <= 32 and LinearFunctionEnv 8 < N <= 64 without add_inverts, and always for LinearFunctionEnv N <= 8 /
 * PermutationEnv N <= 16: the reset's workgroups (the finished envs' scrambles) and the step's workgroups (every other env) share the grid, and an
 * env that was reset takes its first step on the wave / lane that finished its scramble.  Everything else -- and the first call of a session,
 * whose list has to be compacted from the flags -- is the two calls; so is a configuration whose episodes are so short that a large share of
 * the batch finishes in every step (there the two launches are as fast or faster).  qg_plan_query(QG_PLAN_RESET_DONE_STEP) says which. */
int qg_vec_reset_done_step(qg_vec *v, uint64_t reset_seed, const void *actions_dev, int action_dtype, const uint8_t *coins_dev, float *rewards_dev,
                           uint8_t *dones_dev, void *stream);
/* Capturing these calls into a caller's hipGraph: once qg_vec_reset_done is in use on a handle, a single qg_vec_step (and the sampling +
 * step calls) leaves the list of the envs it finished for the reset that follows, so that reset needs no compaction launch.  The list
 * lives on the device and every launch that appends to it expects it empty; what the host believes about it only holds within one
 * "session" -- one stream capture, or eager execution on a handle none of whose list launches was ever captured.  The library
 * therefore (a) zeroes the list (a captured memset) before the first appending launch of every capture and compacts the `done` flags
 * itself at the first qg_vec_reset_done of every capture, (b) after any capture, trusts no list in eager calls (they compact every
 * time), and (c) bounds every append by the batch size on the device.  Any order of replays, eager steps and eager resets gives the
 * results of the same calls made eagerly; the cheapest graph is the one that captures step and reset_done together. */
/* Device clock for launches that are replayed from a captured hipGraph (kernel arguments, hence
 * seeds and RNG counters, are baked into a graph).  While set, every kernel of this handle adds
 * *clock_dev to its RNG counter: reset / reset_done draw with seed + 0x9E3779B9 * clock, the
 * add_inverts coin and the PauliEnv permutation draw use counter + clock.  The owner of the graph
 * advances the clock (a device-side add inside the graph) between replays.  NULL detaches.  The
 * word must stay valid while attached.  Synchronises and drops cached rollout graphs. */
int qg_vec_set_clock(qg_vec *v, const uint64_t *clock_dev);
/* Stream ordering without a system-scope fence: everything enqueued on `waiter_stream` after this call
 * runs after everything enqueued on `producer_stream` before it (hipEventRecord + hipStreamWaitEvent with
 * hipEventDisableTiming | hipEventDisableSystemFence).  For side-stream work next to the step stream
 * (the multi-GPU observation all-gather). */
int qg_stream_wait_stream(void *waiter_stream, void *producer_stream);
/* The handle's host-side RNG counters: the next step's add_inverts coin is drawn with counter
 * `step_index` (it advances by one per step) and the next PauliEnv observe() permutation with
 * `observe_index`.  A caller that replays graphs sets them to the position inside the graph so
 * that eager and replayed launches draw alike (effective counter = this + the device clock). */
int qg_vec_set_counters(qg_vec *v, uint64_t step_index, uint64_t observe_index);
/* Seed of the handle's own counter-RNG streams: the add_inverts coin (clifford.rs:266 `gen_bool(0.5)`) and PauliEnv's
 * observe() permutation draw (pauli.rs:657-662), when they are not supplied by the caller.  Handles start with one fixed
 * seed (reproducible runs); a host that wants the reference's thread_rng behaviour -- independent streams per env object
 * and per clone -- gives every handle its own seed. */
int qg_vec_set_seed(qg_vec *v, uint64_t seed);
/* Global index of this handle's env 0 in every counter-RNG draw (reset scramble, coins, PauliEnv targets and permutations):
 * env e draws as env `first_env + e`.  A handle that holds the shard [first_env, first_env + batch) of a larger batch --
 * one rank of a multi-GPU job, SURVEY 8e: GPU g owns envs [g*B/G, (g+1)*B/G) -- then draws exactly what the unsharded
 * batch would, so a sharded run is bit-identical to the single-GPU run of the whole batch.  Default 0. */
int qg_vec_set_env_base(qg_vec *v, uint64_t first_env);
uint64_t qg_vec_get_env_base(const qg_vec *v);
/* Same as qg_vec_reset, with the draws supplied: actions_dev[t*B + e], t < n_draws (int32). */
int qg_vec_reset_with(qg_vec *v, const int32_t *actions_dev, size_t n_draws, void *stream);

/* Env::step for every env (clifford.rs:321-347): one kernel launch.
 *   actions_dev[B]  element type `action_dtype`; out-of-range (incl. negative) = "no gate" but
 *                   depth still decrements (clifford.rs:324,342)
 *   coins_dev[B]    the gen_bool(0.5) draws of maybe_random_invert (clifford.rs:266); NULL = use
 *                   the handle's counter RNG when add_inverts is set; ignored otherwise
 * Results land in the resident reward/done/success/depth buffers (qg_vec_info). */
int qg_vec_step(qg_vec *v, const void *actions_dev, int action_dtype, const uint8_t *coins_dev, void *stream);

/* The same step with HOST buffers (batched flavour, host-pointer variant): actions_host[B] (and coins_host[B] or NULL) are
 * copied to the device, the step runs, and reward / is_final / success are copied back into rewards_host f32[B], dones_host u8[B],
 * success_host u8[B] (each may be NULL) -- all enqueued on `stream` with hipMemcpyAsync, nothing synchronises: with pinned buffers
 * (hipHostMalloc / hipHostRegister) the call returns at once and the outputs are valid after qg_vec_sync(v, stream); pageable buffers
 * work too (the runtime then stages the copies).  The input buffers must stay untouched until the stream has passed the call.
 * When every buffer passed is pinned AND device-mapped (the default of hipHostMalloc; hipHostRegister with hipHostRegisterMapped) and the
 * output buffers are 16-byte aligned, no copy is made at all: the step kernel reads actions / coins in host memory and one kernel writes
 * the outputs there (same contract, a quarter of the time: tools/bench_step_host.py). */
int qg_vec_step_host(qg_vec *v, const void *actions_host, int action_dtype, const uint8_t *coins_host, float *rewards_host, uint8_t *dones_host,
                     uint8_t *success_host, void *stream);

/* T consecutive steps.  actions_dev[t*B + e]; optional per-step outputs rewards_dev[t*B + e],
 * dones_dev[t*B + e] (NULL = only the final values in the resident buffers).
 *   fused == 0: T single-step launches replayed from a cached hipGraph (same results and same
 *               memory traffic as T qg_vec_step calls, without T host launches)
 *   fused != 0: one launch that keeps each env's state in registers across the T steps */
int qg_vec_rollout(qg_vec *v, const void *actions_dev, int action_dtype, size_t n_steps, const uint8_t *coins_dev,
                   float *rewards_dev, uint8_t *dones_dev, int fused, void *stream);

/* Graph rollout whose step t reads the action slice t % period of actions_dev[period][B]: n_steps
 * single-step launches over a small ring of action buffers -- the access pattern of a policy that
 * rewrites one resident action buffer every step (cache-hot), where qg_vec_rollout streams a
 * [n_steps][B] tensor from HBM. */
int qg_vec_rollout_ring(qg_vec *v, const void *actions_dev, int action_dtype, size_t n_steps, size_t period, void *stream);

/* Env::observe for every env, densified the way the Gym adapter does it (adapters.py:50-54):
 * out_dev[e * obs_rows*obs_cols + r*obs_cols + c] in {0,1}, int8. */
int qg_vec_observe_dense(qg_vec *v, int8_t *out_dev, void *stream);
/* Keep the dense observation RESIDENT: from this call on, dense_dev[B * obs_rows * obs_cols] (caller-owned device memory, 16-byte aligned,
 * alive until the handle is destroyed or the tracking is detached with dense_dev == NULL) always holds what qg_vec_observe_dense would
 * write, in stream order after every call that changes states.  What the Gym adapter does after every step (adapters.py:62-72 calls
 * `_full_obs`, :50-54) costs a full D * D-byte rewrite per env; a gate changes the rows of at most two qubits, so the one-step kernel
 * rewrites just those rows (<= 4 rows of D bytes: 128 of CliffordEnv 16q's 1 024 bytes) in the launch that applies the gate.  Resets of
 * finished envs (qg_vec_reset_done) rewrite those envs; launches that cannot track it (add_inverts, fused rollouts, the policy kernels)
 * are followed by one full rewrite on the same stream.  Handles whose matrix has 16 or 32 rows held as 32-bit words (CliffordEnv
 * N = 8, 16; LinearFunctionEnv N = 16, 32): QG_ERR_UNSUPPORTED otherwise.  Drop cached expectations: rollout graphs are keyed by it. */
int qg_vec_track_dense(qg_vec *v, int8_t *dense_dev, void *stream);
/* Bit-packed observation in QG_FMT_PACKED layout (what the multi-GPU all-gather moves): qg_vec_info.packed_words_per_env
 * words of packed_word_bytes per env, one word per observation row, bit c = column c.  PauliEnv (thread-per-env family,
 * at most 64 observation columns): one 64-bit word per row of the [2N, 2N + max_rotations] observation; the call counts
 * as one Env::observe, i.e. it draws the add_perms permutation like qg_vec_observe_dense does. */
int qg_vec_observe_packed(qg_vec *v, void *out_dev, void *stream);
/* Host-pointer variants of the two calls above: the observation is written to a buffer of the handle and copied to out_host on `stream`
 * (hipMemcpyAsync; valid after qg_vec_sync(v, stream), pinned memory for a truly asynchronous copy).  Sizes: B * obs_rows * obs_cols bytes,
 * resp. B * packed_words_per_env * packed_word_bytes. */
int qg_vec_observe_dense_host(qg_vec *v, int8_t *out_host, void *stream);
int qg_vec_observe_packed_host(qg_vec *v, void *out_host, void *stream);
/* Env::masks (clifford.rs:349-351): out_dev[e*num_actions + a] = !success[e] */
int qg_vec_masks(qg_vec *v, uint8_t *out_dev, void *stream);

/* PauliEnv: rebuild every env from an explicit target, the deterministic tail of
 * PauliEnv::reset (pauli.rs:573-585).  tableaus: [B, 2N*2N] uint8 (host), labels: per env
 * `n_rot[e]` Pauli labels of N characters each, concatenated without separators. */
int qg_vec_pauli_reset_from(qg_vec *v, const uint8_t *tableaus, const char *labels, const int32_t *n_rot, void *stream);

/* PauliEnv with add_perms: observe() draws one of the coupling map's qubit automorphisms per env,
 * permutes the observation with it and remembers it so that the next step() un-permutes the
 * action (pauli.rs:594-599, 653-665).  qg_vec_observe_dense draws from the handle's counter RNG;
 * this variant takes the draws explicitly: perm_idx_dev[B] (int32, reduced mod the number of
 * perms), or NULL for the RNG. */
int qg_vec_pauli_observe_dense(qg_vec *v, int8_t *out_dev, const int32_t *perm_idx_dev, void *stream);
/* number of (qubit, action) permutation pairs of a PauliEnv batch (0 when add_perms is off) */
int qg_vec_pauli_num_perms(const qg_vec *v);

/* Diagnostics: the kernel device clock -- how long a launch's waves were on the machine, measured by the waves themselves (no profiler, no host
 * clock).  slots_dev: n_slots x waves_per_slot records {uint64 entry, uint64 exit} in device memory (16-byte aligned), zeroed by the caller;
 * ticks of the device's constant-rate counter (qg_kernel_clock_rate_khz: 100 MHz on gfx950).  After the call the k-th step / observation kernel
 * launched through this handle (eager, or captured into a graph: a replay stamps the slots its launches were captured with) has its wave w write
 * record w of slot k (waves past waves_per_slot write nothing: size it to the largest grid, batch / 32 covers every stamped kernel); the launch's
 * duration is max(exit) - min(entry) over the records with exit != 0.  Launches past n_slots and kernels without stamps (export and
 * fused-rollout kernels, the 64-bit-row and lane-group resets) leave their slot untouched.  Stamped: the one-step kernels (qm_step1 / q64_step1 /
 * qm_inv2 / q64_inv2 / word_step / lfd_step / ptile_step1c), the resets qm_init / word_init / ptile_generate / ptile_reset_tree (one slot each),
 * qm_reset_step / word_reset_step and the dense observation rewrite (qm_dense_stream).  A wave with a slot waits for its own loads and
 * stores before its exit stamp, so exit - entry covers the memory traffic of the launch; a launch WITHOUT a slot pays one scalar instruction.
 * n_slots = 0 detaches.  Drops cached rollout graphs. */
int qg_vec_set_kernel_clock(qg_vec *v, uint64_t *slots_dev, size_t n_slots, uint32_t waves_per_slot);
/* ticks per millisecond of that counter on `device` (hipDeviceAttributeWallClockRate), or a negative status */
int qg_kernel_clock_rate_khz(int device);

/* Blocks until everything enqueued on `stream` by this handle has finished; returns
 * QG_ERR_PANIC if any env has a fault bit set. */
int qg_vec_sync(qg_vec *v, void *stream);

/* Solution log (Env::solution, clifford.rs:376-381 / pauli.rs:685-719) of env e, host side.
 * Returns the length (may exceed cap) or a negative status.  Entries are kept as 32-bit words on
 * the device: an out-of-range action the reference would log verbatim (clifford.rs:334-340) is
 * reported exactly when it is below 2^32 - 1; negative or larger ones are reported as
 * UINT64_MAX (what `-1 as usize` is). */
int64_t qg_vec_solution(qg_vec *v, uint64_t env, uint64_t *out, size_t cap);
/* The same for EVERY env with one copy of the log: out[e * cap + i] = entry i of env e's list (entries beyond cap dropped), lens[e] = its
 * full length (what twisterl's collector does per episode with Env::solution, here once per batch).  Host pointers; synchronises. */
int qg_vec_solutions(qg_vec *v, uint64_t *out, size_t cap, int64_t *lens);

/* ------------------------------------------------------------------------------------------
 * Collector support: the device-side steps between observe() and step() when a policy network
 * is in the loop.  The reference runs these on the CPU inside twisterl's collector
 * (rl/synthesis.py:128-138; collecting parameters gamma / lambda, rl/configs.py:134-144,219-220).
 * The free functions work on the calling thread's current HIP device.
 * ---------------------------------------------------------------------------------------- */
typedef enum { QG_DT_I8 = 0, QG_DT_F32 = 1, QG_DT_BF16 = 2, QG_DT_F16 = 3 } qg_dtype;

/* Env::observe densified (adapters.py:50-54) straight into the dtype the policy network reads:
 * out_dev[e * obs_rows*obs_cols + r*obs_cols + c] in {0, 1} as `out_dtype`. */
int qg_vec_observe_dense_as(qg_vec *v, void *out_dev, int out_dtype, void *stream);
/* Dense {0,1} tensor from QG_FMT_PACKED rows (e.g. the gathered observation of other ranks, or
 * the packed observations a rollout buffer keeps): n_rows words of `word_bytes` (4 / 8: bit c =
 * column c; 1: PermutationEnv, the byte is the set column) -> out_dev[row * cols + c]. */
int qg_expand_packed(const void *packed_dev, int word_bytes, uint64_t n_rows, uint32_t cols, void *out_dev, int out_dtype, void *stream);
/* The int8 {0,1} observation of qg_vec_observe_dense (n_elems entries) in another dtype -- for
 * a PauliEnv observation taken once (observe() may draw a permutation) that has no row-word form, so it
 * is taken once and widened. */
int qg_widen_dense(const int8_t *obs_dev, uint64_t n_elems, void *out_dev, int out_dtype, void *stream);
/* One categorical draw per env from softmax(logits[e, 0:num_actions]) by an exponential race on
 * the counter RNG: base = rng(seed ^ 0x73616D70, e, counter + *clock_dev) (clock_dev may be NULL);
 * u[a] = ((hash32(base, a) >> 9) + 0.5) *
 * 2^-23; action = argmin -log(u[a]) / exp(logit[a] - max) (kernels_collect.hip states hash32).
 * logits_dev: [batch, ld] of `logits_dtype` (f32 / bf16 / f16), ld >= num_actions.  mask_dev:
 * [batch, num_actions] (1 = allowed; Env::masks) or NULL.  Outputs (each may be NULL except
 * actions): action, log-prob of it, entropy of the row, and values_dev[e] = logits[e, value_col]
 * (value_col >= 0: a value head computed by the same GEMM as the logits). */
int qg_sample_actions(const void *logits_dev, int logits_dtype, uint64_t ld, uint64_t batch, uint32_t num_actions, const uint8_t *mask_dev,
                      uint64_t seed, uint64_t counter, const uint64_t *clock_dev, void *actions_dev, int action_dtype, float *logp_dev,
                      float *entropy_dev, int32_t value_col, float *values_dev, void *stream);
/* Generalised advantage estimation over a [n_steps, batch] rollout (f32, done_t = the episode ended
 * with step t): delta_t = r_t + gamma*V_{t+1}*(1-done_t) - V_t, A_t = delta_t +
 * gamma*lambda*(1-done_t)*A_{t+1}, returns = A + V.  last_values_dev: V after the last step
 * (NULL = 0); returns_dev may be NULL. */
int qg_gae(const float *rewards_dev, const float *values_dev, const uint8_t *dones_dev, const float *last_values_dev, float gamma,
           float gae_lambda, size_t n_steps, uint64_t batch, float *advantages_dev, float *returns_dev, void *stream);

/* The policy's first layer computed straight from the resident bit-packed state (no dense observation
 * is written or read): out[e, n] = act(sum_k obs[e, k] * W[n, k] + bias[n]) in bf16, obs = Env::observe
 * densified and flattened as in qg_vec_observe_dense_as -- the Linear(prod(obs_shape) -> hidden) that opens
 * the reference's policy network (rl/configs.py:531-607; examples/models/ *.pt).  Handles whose rows are resident uint32
 * words only (CliffordEnv N <= 16, LinearFunctionEnv 8 < N <= 32 with or without add_inverts; QG_ERR_UNSUPPORTED otherwise);
 * hidden % 64 == 0.
 * qg_vec_pack_embedding re-orders W ([hidden, ld] of f32 / bf16, ld >= obs_rows*obs_cols) into the k order
 * the kernel's in-register bit expansion produces (qg_vec_embed_packed_bytes bytes; repeat after every
 * optimiser step); qg_vec_embed runs the layer: bias_dev f32 [hidden] or NULL, relu != 0 applies max(0, .),
 * out_dev bf16 [batch, ld_out], 16-byte aligned, ld_out >= hidden and a multiple of 8. */
size_t qg_vec_embed_packed_bytes(const qg_vec *v, uint32_t hidden);
int qg_vec_pack_embedding(qg_vec *v, const void *weight_dev, int weight_dtype, uint64_t ld, uint32_t hidden, void *packed_dev, void *stream);
int qg_vec_embed(qg_vec *v, const void *packed_dev, const float *bias_dev, uint32_t hidden, int relu, void *out_dev, uint64_t ld_out, void *stream);
/* qg_vec_observe_packed(v, obs_packed_dev, stream) and qg_vec_embed(...) of the same state: the observation a rollout stores and the
 * first layer of the policy that acts on it.  One launch for small batches (the layer's kernel holds the bits it would export), the
 * two launches otherwise; results are those of the two calls. */
int qg_vec_embed_observe(qg_vec *v, const void *packed_dev, const float *bias_dev, uint32_t hidden, int relu, void *out_dev, uint64_t ld_out,
                         void *obs_packed_dev, void *stream);

/* The same first layer from packed observation WORDS instead of a handle's resident state: words_dev = [batch, rows] uint64,
 * bit c of word r = observation entry (r, c) -- what qg_vec_observe_packed writes for handles with 64-bit row words (PauliEnv:
 * 2N rows of 2N + max_rotations columns; CliffordEnv N > 16; LinearFunctionEnv N > 32), a packed rollout buffer, or another
 * rank's all-gathered shard, so the learner side never unpacks either.  rows even, cols <= 64, hidden % 128 == 0
 * (qg_policy_embed_words_packed_bytes returns 0 otherwise); W is [hidden, ld] with ld >= rows*cols, k = r*cols + c as in
 * qg_vec_observe_dense_as; words 16-byte aligned; output as qg_vec_embed.  The weights stream through LDS, so K is not limited. */
size_t qg_policy_embed_words_packed_bytes(uint32_t rows, uint32_t cols, uint32_t hidden);
int qg_policy_pack_embed_words(const void *weight_dev, int weight_dtype, uint64_t ld, uint32_t rows, uint32_t cols, uint32_t hidden, void *packed_dev,
                               void *stream);
int qg_policy_embed_words(const uint64_t *words_dev, uint64_t batch, uint32_t rows, uint32_t cols, const void *packed_dev, const float *bias_dev,
                          uint32_t hidden, int relu, void *out_dev, uint64_t ld_out, void *stream);

/* The policy's last layer and the categorical draw in one kernel: logits[e, a] = sum_k h[e, k] W[a, k] + b[a] stay in
 * registers and go straight into the exponential race of qg_sample_actions (same counter RNG, hash and tie rule, so
 * the same seed / counter / clock give the same draw for the same logits); outputs as there, values_dev[e] = the value
 * head W[value_row] . h[e] + b[value_row].  h_dev: bf16 [batch, ld_h], 16-byte aligned, ld_h % 8 == 0.  Limits
 * (qg_policy_head_packed_bytes returns 0 outside them): num_actions <= 222, in_features % 64 == 0, <= 512, packed head
 * <= 144 KiB.  qg_policy_pack_head re-orders W ([rows, ld] f32 / bf16) and the bias ([rows], same dtype, or NULL)
 * into MFMA fragment order: rows 0..num_actions-1 are the actions, row value_row (>= 0, any index) the value head. */
size_t qg_policy_head_packed_bytes(uint32_t num_actions, uint32_t in_features);
/* after_mid != 0: pack for qg_policy_mid_head_sample (whose head fragments come out of an accumulator tile in a permuted k order) */
int qg_policy_pack_head(const void *weight_dev, const void *bias_dev, int dtype, uint64_t ld, uint32_t in_features, uint32_t num_actions,
                        int32_t value_row, int after_mid, void *packed_dev, void *stream);
int qg_policy_head_sample(const void *h_dev, uint64_t ld_h, uint64_t batch, uint32_t in_features, const void *packed_dev, uint32_t num_actions,
                          uint64_t seed, uint64_t counter, const uint64_t *clock_dev, void *actions_dev, int action_dtype, float *logp_dev,
                          float *entropy_dev, float *values_dev, void *stream);
/* The layer before the head as well: h2 = relu(h W2^T + b2) (mid_features = 256 outputs, in_features % 32 == 0, <= 2048) feeds the head
 * from registers -- with qg_vec_embed in front, the reference's default policy (rl/configs.py:531-607) runs forward and samples
 * without a tensor library and without writing an observation, h2 or logits.  qg_policy_pack_mid packs W2 [mid_features, ld] / b2;
 * the head must be packed with after_mid = 1. */
size_t qg_policy_mid_packed_bytes(uint32_t in_features, uint32_t mid_features);
int qg_policy_pack_mid(const void *weight_dev, const void *bias_dev, int dtype, uint64_t ld, uint32_t in_features, uint32_t mid_features,
                       void *packed_dev, void *stream);
int qg_policy_mid_head_sample(const void *h_dev, uint64_t ld_h, uint64_t batch, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                              const void *packed_head_dev, uint32_t num_actions, uint64_t seed, uint64_t counter, const uint64_t *clock_dev, void *actions_dev,
                              int action_dtype, float *logp_dev, float *entropy_dev, float *values_dev, void *stream);
/* The same kernel followed, in the same launch, by Env::step of handle `v` with the action it drew (the lane that holds an env's draw
 * gathers / scatters that env's rows like qg_vec_step does: same results, one launch less per collection step), and by the compaction of
 * the envs whose episode ended with this step: the next qg_vec_reset_done finds its list ready and launches only the reset itself.
 * batch, num_actions, the device clock and the env outputs are the handle's; rewards_dev / dones_dev: per-step outputs as in
 * qg_vec_rollout (may be NULL).  TILE-layout handles (CliffordEnv N <= 16 with or without add_inverts -- the coin is the handle's counter
 * RNG, as in qg_vec_step without coins; LinearFunctionEnv 8 < N <= 32 without add_inverts); QG_ERR_UNSUPPORTED otherwise.  Where the step
 * cannot ride in the sampling kernel (add_inverts beyond the small-batch kernel, or a state that is not symplectic) the call issues
 * the env's own step launch after it: same results. */
int qg_vec_mid_head_sample_step(qg_vec *v, const void *h_dev, uint64_t ld_h, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                                const void *packed_head_dev, uint64_t seed, uint64_t counter, void *actions_dev, int action_dtype, float *logp_dev,
                                float *entropy_dev, float *values_dev, float *rewards_dev, uint8_t *dones_dev, void *stream);
/* ... and by qg_vec_reset_done(v, reset_seed): the envs whose episode ended with this step start their next one (clifford.rs:306-318)
 * before the call returns to the stream -- inside the same launch for small batches, as the reset's own launch otherwise; results are
 * those of the two calls.  A collection loop that uses it needs no qg_vec_reset_done of its own after the first step. */
int qg_vec_mid_head_sample_step_reset(qg_vec *v, const void *h_dev, uint64_t ld_h, uint32_t in_features, const void *packed_mid_dev, uint32_t mid_features,
                                      const void *packed_head_dev, uint64_t seed, uint64_t counter, void *actions_dev, int action_dtype, float *logp_dev,
                                      float *entropy_dev, float *values_dev, float *rewards_dev, uint8_t *dones_dev, uint64_t reset_seed, void *stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU hand-over (BASELINE.json north_star: "batches shard trivially over the 8 GPUs of one node with an RCCL all-gather
 * over xGMI only for the observation tensor handed back to the learner").  One process (or thread) per GPU; rank r
 * of W owns the contiguous env range [r*B, (r+1)*B) of one batch of W*B envs (qg_vec_set_env_base makes its draws those of the
 * whole batch).  Env::step (clifford.rs:321-347) touches one env only, so stepping needs no exchange; what crosses GPUs is what
 * the reference's collector reads from every env after a step -- Env::observe (clifford.rs:361-368), Env::reward (:355),
 * Env::is_final (:353), Env::success (:357-359) -- as ONE flat shard per rank:
 *
 *   [ obs: B * packed_words_per_env * packed_word_bytes (qg_vec_observe_packed) | pad to 4 | reward: f32[B] |
 *     is_final: u8[B] | pad to 4 | success: u8[B] | pad to 16 ]
 *
 * Two transports: ncclAllGather of librccl.so.1 (loaded on the first qg_comm_init, not a link-time dependency), and a direct
 * write in which every rank copies its shard into a window in each peer's HBM over xGMI (hipIpc-shared, one link per peer) and
 * raises a per-source flag -- no collective library on the data path.  A gathered buffer is W consecutive shards, rank order =
 * env order.
 * ---------------------------------------------------------------------------------------- */
typedef struct qg_comm qg_comm;
#define QG_COMM_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */
#define QG_P2P_HANDLE_BYTES 64 /* = HIP_IPC_HANDLE_SIZE */
#define QG_COMM_MAX_WORLD 16

typedef struct {
    uint64_t batch;          /* envs in the shard */
    uint64_t bytes;          /* size of one rank's shard (multiple of 16) = stride between ranks in a gathered buffer */
    uint64_t obs_offset;     /* 0 */
    uint64_t obs_bytes;      /* batch * packed_words_per_env * packed_word_bytes */
    uint64_t reward_offset;  /* f32[batch] */
    uint64_t final_offset;   /* u8[batch]  Env::is_final */
    uint64_t success_offset; /* u8[batch]  Env::success */
} qg_shard_layout;

/* Layout of one rank's shard for this handle (same for every rank that holds the same env kind and batch). */
int qg_vec_learner_shard_layout(const qg_vec *v, qg_shard_layout *out);
/* Write this handle's shard to shard_dev (qg_shard_layout.bytes, 16-byte aligned): two launches on `stream` (the packed
 * observation as qg_vec_observe_packed writes it -- for a PauliEnv this counts as one observe() -- and the three per-env arrays). */
int qg_vec_pack_learner_shard(qg_vec *v, void *shard_dev, void *stream);

/* Rank 0 calls qg_comm_unique_id and hands the 128 bytes to every rank by whatever channel the host has (ncclGetUniqueId). */
int qg_comm_unique_id(uint8_t id_out[QG_COMM_ID_BYTES]);
/* Collective over all ranks: ncclCommInitRank on GPU `device`.  world == 1 is allowed. */
int qg_comm_init(const uint8_t id[QG_COMM_ID_BYTES], int rank, int world, int device, qg_comm **out);
/* A communicator without RCCL, for the direct-write transport only (handles are exchanged by the host: qg_comm_p2p_export /
 * qg_comm_p2p_open).  Not collective. */
int qg_comm_init_local(int rank, int world, int device, qg_comm **out);
/* With the direct-write transport, call it only when every rank is done with every window (a barrier of the host's own): peers write into this
 * rank's window and this rank holds mappings of theirs. */
void qg_comm_destroy(qg_comm *c);
int qg_comm_rank(const qg_comm *c);
int qg_comm_world(const qg_comm *c);

/* qg_vec_pack_learner_shard into a buffer of the communicator, then ncclAllGather of it into out_dev
 * (world * qg_shard_layout.bytes), both on `stream`: stream-ordered after the steps enqueued before it. */
int qg_vec_gather_learner_shard(qg_vec *v, qg_comm *c, void *out_dev, void *stream);
/* The same, double buffered and overlapped with what `stream` does next: submit snapshots the shard on `stream` (the pack
 * launches) and hands the PREVIOUS snapshot to ncclAllGather on the communicator's own stream; flush hands over the last one.  The
 * hand-over goes through the host (wait for the snapshot's event, then enqueue the collective), not through a stream-to-stream
 * event wait, which on ROCm 7 / MI355X slows every later hipGraph replay on the stepping stream by ~40 % (tools/gather_probe.py);
 * the host therefore runs at most one snapshot ahead of the device.  latest waits for the most recently handed-over collective
 * and returns its buffer (owned by the communicator, valid until two more submits), or NULL in *gathered_dev when none exists. */
int qg_comm_gather_submit(qg_comm *c, qg_vec *v, void *stream);
int qg_comm_gather_flush(qg_comm *c);
int qg_comm_gather_latest(qg_comm *c, const void **gathered_dev);

/* Direct write.  Set-up, once per communicator: every rank allocates its window (header + 2 parities x world x shard_bytes of
 * uncached device memory), the hipIpc handles are exchanged, every rank maps the others' windows.  qg_comm_p2p_connect does all
 * three over the communicator's RCCL (collective); without RCCL the host calls qg_comm_p2p_export on every rank, moves the
 * QG_P2P_HANDLE_BYTES to every other rank itself (rank order) and calls qg_comm_p2p_open (a rank never opens its own handle). */
int qg_comm_p2p_connect(qg_comm *c, uint64_t shard_bytes);
int qg_comm_p2p_export(qg_comm *c, uint64_t shard_bytes, uint8_t handle_out[QG_P2P_HANDLE_BYTES]);
int qg_comm_p2p_open(qg_comm *c, const uint8_t *handles /* [world][QG_P2P_HANDLE_BYTES] */);
/* Epoch k (1, 2, ...; the k-th call on every rank): pack this handle's shard, copy it into slot `rank` of parity k & 1 of every
 * rank's window (own included) and store k to the per-source arrival flag there; all on `stream`.  Before overwriting a parity
 * the kernel waits (bounded) for that peer's release of epoch k - 2. */
int qg_vec_push_learner_shard(qg_vec *v, qg_comm *c, void *stream);
/* Enqueue on `stream` the wait for every rank's epoch-k arrival in this rank's window (k = the k-th call) and return the
 * gathered buffer of that epoch (world * shard_bytes, inside the window): work enqueued on `stream` afterwards may read it. */
int qg_comm_p2p_wait(qg_comm *c, const void **gathered_dev, void *stream);
/* Enqueue on `stream` the release of the epoch last waited for: its parity may be overwritten by the push of epoch k + 2. */
int qg_comm_p2p_release(qg_comm *c, void *stream);
/* Synchronise `stream` and report a peer that did not arrive / release within the deadline (QG_ERR_DEVICE), else QG_OK.
 * Deadlines: a push whose target parity was not released in time SKIPS that peer (no copy, no arrival flag: the peer's wait runs into its
 * own deadline rather than reading a torn shard) and raises a sticky error; while it is set, pushes write to nobody.  qg_comm_p2p_wait
 * returns its pointer either way: a caller that may have lost a peer checks before it trusts an epoch. */
int qg_comm_p2p_check(qg_comm *c, void *stream);
/* Restart the direct-write transport after an error: call on EVERY rank between two barriers of the host's own (all streams drained, no
 * push in flight anywhere).  Synchronises `stream`, clears the error word, this rank's window header and both epoch counters. */
int qg_comm_p2p_reset(qg_comm *c, void *stream);

/* ------------------------------------------------------------------------------------------
 * Which kernel would run.  The library picks a state layout and a kernel family from the env kind, the sizes and the options (DESIGN.md
 * section 2); this query answers from the same decision functions the launch paths call (csrc/qgym_plan.hpp) and needs no GPU, so a test can
 * pin the table.  `arg`: QG_PLAN_ROLLOUT_FUSED: number of steps T; QG_PLAN_RESET_DONE: length of the list of finished envs.
 * `nonsymplectic` != 0: some env holds a state not known to be symplectic (set_state of an arbitrary matrix with add_inverts).
 * name_out receives a short name ("TILE", "qm_step1_kernel", "scramble_tree", "qm_dense_stream_kernel", ...).  Returns QG_OK, or the status
 * qg_vec_create would return for this configuration (QG_ERR_UNSUPPORTED beyond the limits).
 * ---------------------------------------------------------------------------------------- */
typedef enum {
    QG_PLAN_LAYOUT = 0,         /* the resident state layout */
    QG_PLAN_STEP = 1,           /* qg_vec_step */
    QG_PLAN_ROLLOUT_FUSED = 2,  /* qg_vec_rollout(fused = 1) of `arg` steps */
    QG_PLAN_RESET_DONE = 3,     /* qg_vec_reset_done with `arg` finished envs, cfg->difficulty draws each */
    QG_PLAN_OBSERVE_DENSE = 4,  /* qg_vec_observe_dense into a 16-byte-aligned buffer */
    QG_PLAN_OBSERVE_PACKED = 5, /* qg_vec_observe_packed */
    QG_PLAN_STATE_I64 = 6,      /* qg_vec_get_state / set_state in QG_FMT_I64 */
    QG_PLAN_TRACK_DENSE = 7,    /* qg_vec_track_dense: "in-step", "refresh" (a full rewrite after every step) or unsupported */
    QG_PLAN_RESET_DONE_STEP = 8 /* qg_vec_reset_done_step: the one-launch kernel, or "two launches" */
} qg_plan_op;
int qg_plan_query(const qg_config *cfg, uint64_t batch, uint32_t num_actions, int op, uint64_t arg, int nonsymplectic, char *name_out, size_t cap);

/* ------------------------------------------------------------------------------------------
 * Scalar environment: the `Env` trait method for method (clifford.rs:285-382): a batch of one on the same kernels.  A call that changes the
 * state costs one kernel launch and one stream synchronisation where the layout keeps a resident dense observation (16- / 32-row matrices: the
 * step kernel rewrites the changed rows of the pinned observation itself), one more launch elsewhere; results and the fault word land in a pinned
 * block the getters read as plain loads.  This flavour exists for API parity -- throughput comes from qg_vec_* (README: 5.7e4 env-steps/s per host
 * thread, 1.9e5 on 32 threads -- the HIP runtime's own launch rate -- against 1e10 batched).
 * qg_env_destroy parks the handle (device buffers, stream, pinned block) in a process-wide pool for the next qg_env_clone of an env with
 * the same constructor arguments -- at most 64 per configuration, 256 in all; beyond that it frees.  qg_env_pool_clear releases the pool
 * (call it at teardown, or when a configuration will not be cloned again).
 * ---------------------------------------------------------------------------------------- */
typedef struct qg_env qg_env;

int qg_env_create(const qg_config *cfg, const qg_gate *gates, size_t n_gates, int device, qg_env **out);
int qg_env_clone(const qg_env *e, qg_env **out); /* Env: DynClone */
/* seed of the env's own RNG streams (see qg_vec_set_seed); a clone starts with its source's seed and counters */
int qg_env_set_seed(qg_env *e, uint64_t seed);
void qg_env_destroy(qg_env *e);
void qg_env_pool_clear(void);
int64_t qg_env_num_actions(const qg_env *e);
int qg_env_obs_shape(const qg_env *e, int64_t out[2]);
int qg_env_set_difficulty(qg_env *e, int64_t d);
int64_t qg_env_get_difficulty(const qg_env *e);
int qg_env_set_state(qg_env *e, const int64_t *state, size_t n);
int qg_env_reset(qg_env *e, uint64_t seed);
int qg_env_step(qg_env *e, int64_t action);
/* step with the inversion coin supplied (parity runs with add_inverts) */
int qg_env_step_coin(qg_env *e, int64_t action, int coin);
int64_t qg_env_masks(const qg_env *e, uint8_t *out, size_t cap);
int qg_env_is_final(const qg_env *e);
float qg_env_reward(const qg_env *e);
int qg_env_success(const qg_env *e);
/* ascending flat indices of set observation entries; returns the count */
int64_t qg_env_observe(qg_env *e, int64_t *out, size_t cap);
int qg_env_track_solution(const qg_env *e);
int64_t qg_env_solution(const qg_env *e, uint64_t *out, size_t cap);
/* twists(): (obs_perms, act_perms) (clifford.rs:370-372).  Returns the number of twists;
 * obs_perms_out[n * obs_size], act_perms_out[n * num_actions] when non-NULL. */
int64_t qg_env_twists(const qg_env *e, int64_t *obs_perms_out, int64_t *act_perms_out);

#ifdef __cplusplus
}
#endif
#endif /* QGYM_H */
