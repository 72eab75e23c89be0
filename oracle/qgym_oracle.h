/*
 * qgym_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's scalar `env.step()` hot path, written from the
 * Rust text of AI4quantum/qiskit-gym (rust/src/envs/{clifford,linear_function,permutation,
 * pauli,metrics,common}.rs and rust/src/pauli/{pauli,pauli_dag,pauli_network}.rs).  It keeps the
 * reference's data layout (one byte per GF(2) entry, row-major), its operation order and its
 * quirks; every function cites the file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library,
 * and only as the checker / the reported CPU baseline.  Nothing under qiskit_gym_amd/ links,
 * imports or calls it.
 *
 * PARITY PINNING: the reference cannot be compiled or imported here (no Rust toolchain, PyO3
 * module absent).  The oracle is pinned against the known-answer data the
 * reference ships (examples/intro.ipynb recorded outputs, examples/models/ JSON, and the three
 * trained policies examples/models/ *.pt) -- see tests/golden/ -- for gateset ordering,
 * LinearFunction/Permutation/Clifford state transitions and is_final; the trained policies, run
 * greedily, solve random targets on this oracle (and on the HIP path) and fail on an env that
 * differs in observation layout, action order or one gate's semantics (tests/test_reference_policies.py).
 * PauliEnv, for which the reference ships no data at all, is pinned to physics through the reference's Python
 * encode / decode contract (envs/synthesis.py:316-512) restated on explicit unitaries: encoded circuits, solved by
 * replaying their gates, decode to circuits with the same unitary (tests/test_physics.py).  Still "parity
 * unpinned": the order of the observation's rotation columns (petgraph 0.6.5 `retain_nodes`/`remove_node`
 * swap-remove re-indexing, restated from the crate's published algorithm) and twisterl 0.5.1's `Env` trait.
 *
 * Randomness: the reference draws from rand::thread_rng() (unseedable).  The oracle takes every
 * random draw as an explicit argument: reset scramble actions, the add_inverts coin, the Pauli
 * observe() permutation index.
 */
#ifndef QGYM_ORACLE_H
#define QGYM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rust/src/envs/common.rs:19-29 */
enum { OG_H = 0, OG_S = 1, OG_SDG = 2, OG_SX = 3, OG_SXDG = 4, OG_CX = 5, OG_CZ = 6, OG_SWAP = 7 };
enum { OG_ENV_PERMUTATION = 0, OG_ENV_LINEAR_FUNCTION = 1, OG_ENV_CLIFFORD = 2, OG_ENV_PAULI = 3 };

typedef struct {
    int32_t kind;
    int32_t q0;
    int32_t q1; /* ignored for 1-qubit gates */
} og_gate;

typedef struct {
    int32_t num_qubits;
    int32_t difficulty;
    int32_t depth_slope;
    int32_t max_depth;
    /* metrics.rs:150-166 */
    float w_n_cnots, w_n_layers_cnots, w_n_layers, w_n_gates;
    int32_t add_inverts;
    int32_t add_perms; /* twists (symmetry.rs) for Env::twists; PauliEnv: its internal qubit / action permutations */
    int32_t track_solution;
    /* PauliEnv only (pauli.rs:340-409) */
    int32_t max_rotations;
    int32_t pauli_diff_scale;
    int32_t final_pauli_layers;
    float num_qubits_decay;
    float pauli_layer_reward;
} og_config;

typedef struct og_env og_env;

/* Fills the reference's constructor defaults (clifford.rs:401-426, pauli.rs:743-778,
 * metrics.rs:157-166, envs/synthesis.py:182-204,380-412). */
void og_config_default(og_config *cfg, int32_t env_kind, int32_t num_qubits);

og_env *og_env_new(int32_t env_kind, const og_config *cfg, const og_gate *gates, size_t n_gates);
og_env *og_env_clone(const og_env *e);
void og_env_free(og_env *e);
const char *og_last_error(void);

/* --- the twisterl::rl::env::Env method set (inferred from clifford.rs:285-382) --- */
size_t og_env_num_actions(const og_env *e);
size_t og_env_obs_shape(const og_env *e, size_t out[2]);
void og_env_set_difficulty(og_env *e, size_t d);
size_t og_env_get_difficulty(const og_env *e);
/* returns 0, or -1 when the reference would panic (malformed Pauli state) */
int og_env_set_state(og_env *e, const int64_t *state, size_t n);
/* reset(): `actions` are the `difficulty` uniform draws of reset() (clifford.rs:311-316).
 * For PauliEnv use og_pauli_reset_from(). Returns -1 if n != difficulty. */
int og_env_reset_with(og_env *e, const int64_t *actions, size_t n);
/* step(action): `coin` is the gen_bool(0.5) of maybe_random_invert (clifford.rs:266); it is
 * only consulted when add_inverts is set.  Returns 0, or -1 where the reference panics. */
int og_env_step(og_env *e, size_t action, int coin);
size_t og_env_masks(const og_env *e, uint8_t *out, size_t cap);
int og_env_is_final(const og_env *e);
float og_env_reward(const og_env *e);
int og_env_success(const og_env *e);
/* observe(): ascending flat indices of set entries.  perm_idx: PauliEnv's random qubit-perm
 * draw (pauli.rs:660); ignored by the other envs and when no perms are installed. */
size_t og_env_observe(og_env *e, int64_t *out, size_t cap, size_t perm_idx);
int og_env_track_solution(const og_env *e);
size_t og_env_solution(const og_env *e, uint64_t *out, size_t cap);

/* twists(): (obs_perms, act_perms) (clifford.rs:370-372; empty when add_perms is off, clifford.rs:218-222; always empty for
 * PauliEnv, pauli.rs:675-679).  Returns the number of twists; obs_out[n][prod(obs_shape)], act_out[n][num_actions] when non-NULL. */
int64_t og_env_twists(const og_env *e, int64_t *obs_out, int64_t *act_out);
/* The constructor-time computations behind it (rust/src/envs/symmetry.rs, restated in qgym_oracle_symmetry.c):
 * compute_twists_square / compute_twists_clifford (symmetry.rs:297-303) by env kind ... */
int64_t og_twists(int32_t env_kind, size_t num_qubits, const og_gate *gates, size_t n_gates, int64_t *obs_out, int64_t *act_out);
/* ... and compute_qubit_perms (symmetry.rs:307-361): qubit_out[n][num_qubits], act_out[n][n_gates] */
int64_t og_qubit_perms(size_t num_qubits, const og_gate *gates, size_t n_gates, int64_t *qubit_out, int64_t *act_out);

/* --- white-box accessors used by the parity tests --- */
size_t og_env_depth(const og_env *e);
int og_env_inverted(const og_env *e);
/* Raw state bytes: Clifford/LF: D*D row-major 0/1; Permutation: N entries (as int64 in
 * og_env_get_state_i64).  Pauli: 2N x 2N tableau only. Returns count written. */
size_t og_env_get_state_i64(const og_env *e, int64_t *out, size_t cap);
/* metrics snapshot (metrics.rs:55-62): n_cnots, n_layers_cnots, n_layers, n_gates */
void og_env_metrics(const og_env *e, size_t out[4]);

/* --- PauliEnv specifics --- */
/* The deterministic tail of PauliEnv::reset (pauli.rs:573-585): rebuild the network from an
 * explicit tableau (row-major 2N x 2N of 0/1) and rotation labels ('\0'-separated, n_rot of
 * them), clean trivial rotations, apply the depth rule, reset internals. */
int og_pauli_reset_from(og_env *e, const uint8_t *tableau, const char *labels, size_t n_rot);
/* The whole of PauliEnv::reset (pauli.rs:554-586) including the random target generator
 * (pauli.rs:54-271), every draw taken from the counter RNG stream of (seed, env_index). */
int og_pauli_reset_seeded(og_env *e, uint64_t seed, uint64_t env_index);
/* Replace the qubit/action permutations the constructor computed (pauli.rs:374-378) by explicit ones. */
int og_pauli_set_perms(og_env *e, const int64_t *qubit_perms, const int64_t *act_perms, size_t n_perms);
/* Active rotations in DAG node order (pauli_network.rs:176-181); returns count. */
size_t og_pauli_active_rotations(const og_env *e, int64_t *out, size_t cap);
/* Per-rotation (x bits, z bits, base_phase, phase()) for rotation k; x/z as N bytes. */
int og_pauli_rotation(const og_env *e, size_t k, uint8_t *x, uint8_t *z, int32_t *base_phase, int32_t *phase);
size_t og_pauli_num_rotations(const og_env *e);

/* --- batched driver: B independent clones stepped side by side (mirrors twisterl's
 * rayon-over-episode-clones data parallelism; used by the parity tests and by bench.py's
 * cpu_baseline leg).  `threads` <= 0 means all cores. --- */
typedef struct og_vec og_vec;
og_vec *og_vec_new(const og_env *proto, size_t batch);
void og_vec_free(og_vec *v);
og_env *og_vec_env(og_vec *v, size_t i);
int og_vec_set_state(og_vec *v, const int64_t *states, size_t per_env);
/* actions[t*B + e] for t in 0..n_steps: scramble each env from identity (reset_with) */
int og_vec_reset_with(og_vec *v, const int64_t *actions, size_t n_steps);
/* actions[B], coins[B] or NULL; outputs may be NULL */
int og_vec_step(og_vec *v, const int32_t *actions, const uint8_t *coins, float *reward,
                uint8_t *success, uint8_t *is_final, int32_t *depth, int threads);
/* dense int8 observation [B, prod(obs_shape)] (envs/adapters.py:50-54) */
int og_vec_observe_dense(og_vec *v, int8_t *out, int threads);
int og_vec_get_state(og_vec *v, int64_t *out, size_t per_env);
/* Test conveniences: Env::reset of the envs selected by `mask` (NULL = all) with its draws taken from the counter RNG the HIP path
 * uses, keyed by the GLOBAL env id (env_ids[i], or env_base + i): `difficulty` x gen_range(0..num_actions) for Clifford / LinearFunction /
 * Permutation (clifford.rs:311-316), og_pauli_reset_seeded for PauliEnv; and Env::solution of every env (lens[i] = full length). */
int og_vec_reset_seeded(og_vec *v, uint64_t seed, uint64_t env_base, const uint64_t *env_ids, const uint8_t *mask, int threads);
int og_vec_pauli_reset_seeded(og_vec *v, uint64_t seed, uint64_t env_base, const uint64_t *env_ids, const uint8_t *mask, int threads);
int og_vec_solutions(og_vec *v, uint64_t *out, size_t cap, int64_t *lens);

#ifdef __cplusplus
}
#endif
#endif
