"""ctypes binding of libqgym.so (the C ABI declared in include/qgym.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C qiskit_gym_amd/csrc` into
`qiskit_gym_amd/lib/libqgym.so`.  There is no fallback: if the library is missing, or no gfx950
device is visible, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libqgym.so")

QG_OK = 0
ENV_KIND = {"permutation": 0, "linear_function": 1, "clifford": 2, "pauli": 3}
FMT_I64, FMT_U8, FMT_PACKED = 0, 1, 2
ACT_I32, ACT_I64 = 0, 1


class QGGate(C.Structure):
    _fields_ = [("kind", C.c_int32), ("q0", C.c_int32), ("q1", C.c_int32)]


class QGConfig(C.Structure):
    _fields_ = [
        ("env_kind", C.c_int32),
        ("num_qubits", C.c_int32),
        ("difficulty", C.c_int32),
        ("depth_slope", C.c_int32),
        ("max_depth", C.c_int32),
        ("w_n_cnots", C.c_float),
        ("w_n_layers_cnots", C.c_float),
        ("w_n_layers", C.c_float),
        ("w_n_gates", C.c_float),
        ("add_inverts", C.c_int32),
        ("add_perms", C.c_int32),
        ("track_solution", C.c_int32),
        ("max_rotations", C.c_int32),
        ("pauli_diff_scale", C.c_int32),
        ("final_pauli_layers", C.c_int32),
        ("num_qubits_decay", C.c_float),
        ("pauli_layer_reward", C.c_float),
    ]


class QGVecInfo(C.Structure):
    _fields_ = [
        ("env_kind", C.c_int32),
        ("num_qubits", C.c_int32),
        ("num_actions", C.c_int32),
        ("obs_rows", C.c_int32),
        ("obs_cols", C.c_int32),
        ("device", C.c_int32),
        ("batch", C.c_uint64),
        ("packed_word_bytes", C.c_uint32),
        ("packed_words_per_env", C.c_uint32),
        ("packed_env_stride_bytes", C.c_uint64),
        ("state_dev", C.c_void_p),
        ("reward_dev", C.c_void_p),
        ("done_dev", C.c_void_p),
        ("success_dev", C.c_void_p),
        ("depth_dev", C.c_void_p),
        ("error_dev", C.c_void_p),
    ]


class QGShardLayout(C.Structure):
    _fields_ = [("batch", C.c_uint64), ("bytes", C.c_uint64), ("obs_offset", C.c_uint64), ("obs_bytes", C.c_uint64),
                ("reward_offset", C.c_uint64), ("final_offset", C.c_uint64), ("success_offset", C.c_uint64)]


QG_COMM_ID_BYTES = 128
QG_P2P_HANDLE_BYTES = 64


class QGymError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libqgym status {status}: {message}")
        self.status = status
        self.message = message


# every symbol include/qgym.h declares (tests check the built library exports all of them)
EXPORTED_SYMBOLS = [
    "qg_config_default", "qg_last_error", "qg_abi_version", "qg_device_count", "qg_gate_parse",
    "qg_vec_create", "qg_vec_destroy", "qg_vec_get_info", "qg_vec_bind_outputs", "qg_vec_set_difficulty", "qg_vec_get_difficulty",
    "qg_vec_set_state", "qg_vec_get_state", "qg_vec_reset", "qg_vec_reset_done", "qg_vec_reset_done_step", "qg_vec_set_clock", "qg_stream_wait_stream", "qg_vec_set_counters", "qg_vec_set_seed", "qg_vec_set_env_base", "qg_vec_get_env_base", "qg_vec_reset_with", "qg_vec_step", "qg_vec_step_host", "qg_vec_rollout", "qg_vec_rollout_ring",
    "qg_vec_observe_dense", "qg_vec_track_dense", "qg_vec_observe_packed", "qg_vec_observe_dense_host", "qg_vec_observe_packed_host", "qg_vec_masks", "qg_vec_pauli_reset_from", "qg_vec_pauli_observe_dense", "qg_vec_pauli_num_perms", "qg_vec_sync",
    "qg_vec_solution",
    "qg_vec_solutions",
    "qg_vec_set_kernel_clock",
    "qg_kernel_clock_rate_khz",
    "qg_vec_observe_dense_as", "qg_expand_packed", "qg_widen_dense", "qg_sample_actions", "qg_gae",
    "qg_vec_embed_packed_bytes", "qg_vec_pack_embedding", "qg_vec_embed", "qg_vec_embed_observe",
    "qg_policy_embed_words_packed_bytes", "qg_policy_pack_embed_words", "qg_policy_embed_words",
    "qg_policy_head_packed_bytes", "qg_policy_pack_head", "qg_policy_head_sample",
    "qg_policy_mid_packed_bytes", "qg_policy_pack_mid", "qg_policy_mid_head_sample", "qg_vec_mid_head_sample_step", "qg_vec_mid_head_sample_step_reset",
    "qg_vec_learner_shard_layout", "qg_vec_pack_learner_shard", "qg_comm_unique_id", "qg_comm_init", "qg_comm_init_local", "qg_comm_destroy",
    "qg_comm_rank", "qg_comm_world", "qg_vec_gather_learner_shard", "qg_comm_gather_submit", "qg_comm_gather_flush", "qg_comm_gather_latest",
    "qg_comm_p2p_connect", "qg_comm_p2p_export", "qg_comm_p2p_open", "qg_vec_push_learner_shard", "qg_comm_p2p_wait", "qg_comm_p2p_release",
    "qg_comm_p2p_check", "qg_comm_p2p_reset", "qg_plan_query",
    "qg_env_create", "qg_env_clone", "qg_env_set_seed", "qg_env_destroy", "qg_env_pool_clear", "qg_env_num_actions", "qg_env_obs_shape",
    "qg_env_set_difficulty", "qg_env_get_difficulty", "qg_env_set_state", "qg_env_reset", "qg_env_step",
    "qg_env_step_coin", "qg_env_masks", "qg_env_is_final", "qg_env_reward", "qg_env_success", "qg_env_observe",
    "qg_env_track_solution", "qg_env_solution", "qg_env_twists",
]

_lib = None


def load():
    """Load libqgym.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        # a fresh checkout on a ROCm box: build the HIP library in-tree (never a fallback path)
        import subprocess

        subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], check=False, capture_output=True)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(qiskit_gym_amd has no CPU fallback)"
        )
    try:  # share torch's HIP runtime when torch is in the process (same libamdhip64.so.7 soname)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, sz, u64, i64 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int64
    L.qg_last_error.restype = C.c_char_p
    L.qg_config_default.argtypes = [C.POINTER(QGConfig), C.c_int32, C.c_int32]
    L.qg_config_default.restype = None
    L.qg_gate_parse.argtypes = [C.c_char_p, C.POINTER(i64), sz, C.POINTER(QGGate)]
    L.qg_vec_create.argtypes = [C.POINTER(QGConfig), C.POINTER(QGGate), sz, u64, C.c_int, C.POINTER(vp)]
    L.qg_vec_destroy.argtypes = [vp]
    L.qg_vec_destroy.restype = None
    L.qg_vec_get_info.argtypes = [vp, C.POINTER(QGVecInfo)]
    L.qg_vec_bind_outputs.argtypes = [vp, vp, vp, vp, vp]
    L.qg_vec_set_difficulty.argtypes = [vp, i64]
    L.qg_vec_get_difficulty.argtypes = [vp]
    L.qg_vec_get_difficulty.restype = i64
    L.qg_vec_set_state.argtypes = [vp, vp, C.c_int, sz, C.c_int, vp]
    L.qg_vec_get_state.argtypes = [vp, vp, C.c_int, sz, C.c_int, vp]
    L.qg_vec_reset.argtypes = [vp, u64, vp]
    L.qg_vec_reset_done.argtypes = [vp, u64, vp]
    L.qg_vec_reset_with.argtypes = [vp, vp, sz, vp]
    L.qg_vec_step.argtypes = [vp, vp, C.c_int, vp, vp]
    L.qg_vec_step_host.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
    L.qg_vec_observe_dense_host.argtypes = [vp, vp, vp]
    L.qg_vec_observe_packed_host.argtypes = [vp, vp, vp]
    L.qg_vec_rollout.argtypes = [vp, vp, C.c_int, sz, vp, vp, vp, C.c_int, vp]
    L.qg_vec_rollout_ring.argtypes = [vp, vp, C.c_int, sz, sz, vp]
    L.qg_vec_observe_dense.argtypes = [vp, vp, vp]
    L.qg_vec_track_dense.argtypes = [vp, vp, vp]
    L.qg_vec_reset_done_step.argtypes = [vp, C.c_uint64, vp, C.c_int, vp, vp, vp, vp]
    L.qg_vec_observe_packed.argtypes = [vp, vp, vp]
    L.qg_vec_masks.argtypes = [vp, vp, vp]
    L.qg_vec_pauli_reset_from.argtypes = [vp, vp, C.c_char_p, vp, vp]
    L.qg_vec_pauli_observe_dense.argtypes = [vp, vp, vp, vp]
    L.qg_vec_pauli_num_perms.argtypes = [vp]
    L.qg_vec_sync.argtypes = [vp, vp]
    L.qg_vec_solution.argtypes = [vp, u64, C.POINTER(u64), sz]
    L.qg_vec_solution.restype = i64
    L.qg_vec_solutions.argtypes = [vp, C.POINTER(u64), sz, C.POINTER(i64)]
    L.qg_vec_set_kernel_clock.argtypes = [vp, vp, sz, C.c_uint32]
    L.qg_kernel_clock_rate_khz.argtypes = [C.c_int]
    L.qg_vec_observe_dense_as.argtypes = [vp, vp, C.c_int, vp]
    L.qg_expand_packed.argtypes = [vp, C.c_int, u64, C.c_uint32, vp, C.c_int, vp]
    L.qg_sample_actions.argtypes = [vp, C.c_int, u64, u64, C.c_uint32, vp, u64, u64, vp, vp, C.c_int, vp, vp, C.c_int32, vp, vp]
    L.qg_widen_dense.argtypes = [vp, u64, vp, C.c_int, vp]
    L.qg_vec_set_clock.argtypes = [vp, vp]
    L.qg_vec_set_counters.argtypes = [vp, u64, u64]
    L.qg_vec_set_seed.argtypes = [vp, u64]
    L.qg_vec_set_env_base.argtypes = [vp, u64]
    L.qg_vec_get_env_base.argtypes = [vp]
    L.qg_vec_get_env_base.restype = u64
    L.qg_env_set_seed.argtypes = [vp, u64]
    L.qg_stream_wait_stream.argtypes = [vp, vp]
    L.qg_gae.argtypes = [vp, vp, vp, vp, C.c_float, C.c_float, sz, u64, vp, vp, vp]
    L.qg_vec_embed_packed_bytes.argtypes = [vp, C.c_uint32]
    L.qg_vec_embed_packed_bytes.restype = sz
    L.qg_vec_pack_embedding.argtypes = [vp, vp, C.c_int, u64, C.c_uint32, vp, vp]
    L.qg_vec_embed.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, u64, vp]
    L.qg_vec_embed_observe.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, u64, vp, vp]
    L.qg_policy_embed_words_packed_bytes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.qg_policy_embed_words_packed_bytes.restype = sz
    L.qg_policy_pack_embed_words.argtypes = [vp, C.c_int, u64, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.qg_policy_embed_words.argtypes = [vp, u64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, C.c_int, vp, u64, vp]
    L.qg_policy_head_packed_bytes.argtypes = [C.c_uint32, C.c_uint32]
    L.qg_policy_head_packed_bytes.restype = sz
    L.qg_policy_pack_head.argtypes = [vp, vp, C.c_int, u64, C.c_uint32, C.c_uint32, C.c_int32, C.c_int, vp, vp]
    L.qg_policy_mid_packed_bytes.argtypes = [C.c_uint32, C.c_uint32]
    L.qg_policy_mid_packed_bytes.restype = sz
    L.qg_policy_pack_mid.argtypes = [vp, vp, C.c_int, u64, C.c_uint32, C.c_uint32, vp, vp]
    L.qg_policy_mid_head_sample.argtypes = [vp, u64, u64, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, u64, u64, vp, vp, C.c_int, vp, vp, vp, vp]
    L.qg_vec_mid_head_sample_step.argtypes = [vp, vp, u64, C.c_uint32, vp, C.c_uint32, vp, u64, u64, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.qg_vec_mid_head_sample_step_reset.argtypes = [vp, vp, u64, C.c_uint32, vp, C.c_uint32, vp, u64, u64, vp, C.c_int, vp, vp, vp, vp, vp, u64, vp]
    L.qg_policy_head_sample.argtypes = [vp, u64, u64, C.c_uint32, vp, C.c_uint32, u64, u64, vp, vp, C.c_int, vp, vp, vp, vp]
    L.qg_vec_learner_shard_layout.argtypes = [vp, C.POINTER(QGShardLayout)]
    L.qg_vec_pack_learner_shard.argtypes = [vp, vp, vp]
    L.qg_comm_unique_id.argtypes = [vp]
    L.qg_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.qg_comm_init_local.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.qg_comm_destroy.argtypes = [vp]
    L.qg_comm_destroy.restype = None
    L.qg_comm_rank.argtypes = [vp]
    L.qg_comm_world.argtypes = [vp]
    L.qg_vec_gather_learner_shard.argtypes = [vp, vp, vp, vp]
    L.qg_comm_gather_submit.argtypes = [vp, vp, vp]
    L.qg_comm_gather_flush.argtypes = [vp]
    L.qg_comm_gather_latest.argtypes = [vp, C.POINTER(vp)]
    L.qg_comm_p2p_connect.argtypes = [vp, u64]
    L.qg_comm_p2p_export.argtypes = [vp, u64, vp]
    L.qg_comm_p2p_open.argtypes = [vp, vp]
    L.qg_vec_push_learner_shard.argtypes = [vp, vp, vp]
    L.qg_comm_p2p_wait.argtypes = [vp, C.POINTER(vp), vp]
    L.qg_comm_p2p_release.argtypes = [vp, vp]
    L.qg_comm_p2p_check.argtypes = [vp, vp]
    L.qg_comm_p2p_reset.argtypes = [vp, vp]
    L.qg_plan_query.argtypes = [C.POINTER(QGConfig), C.c_uint64, C.c_uint32, C.c_int, C.c_uint64, C.c_int, C.c_char_p, C.c_size_t]
    L.qg_env_create.argtypes = [C.POINTER(QGConfig), C.POINTER(QGGate), sz, C.c_int, C.POINTER(vp)]
    L.qg_env_clone.argtypes = [vp, C.POINTER(vp)]
    L.qg_env_destroy.argtypes = [vp]
    L.qg_env_destroy.restype = None
    L.qg_env_pool_clear.argtypes = []
    L.qg_env_pool_clear.restype = None
    L.qg_env_num_actions.argtypes = [vp]
    L.qg_env_num_actions.restype = i64
    L.qg_env_obs_shape.argtypes = [vp, C.POINTER(i64)]
    L.qg_env_set_difficulty.argtypes = [vp, i64]
    L.qg_env_get_difficulty.argtypes = [vp]
    L.qg_env_get_difficulty.restype = i64
    L.qg_env_set_state.argtypes = [vp, C.POINTER(i64), sz]
    L.qg_env_reset.argtypes = [vp, u64]
    L.qg_env_step.argtypes = [vp, i64]
    L.qg_env_step_coin.argtypes = [vp, i64, C.c_int]
    L.qg_env_masks.argtypes = [vp, C.POINTER(C.c_uint8), sz]
    L.qg_env_masks.restype = i64
    L.qg_env_is_final.argtypes = [vp]
    L.qg_env_reward.argtypes = [vp]
    L.qg_env_reward.restype = C.c_float
    L.qg_env_success.argtypes = [vp]
    L.qg_env_observe.argtypes = [vp, C.POINTER(i64), sz]
    L.qg_env_observe.restype = i64
    L.qg_env_track_solution.argtypes = [vp]
    L.qg_env_solution.argtypes = [vp, C.POINTER(u64), sz]
    L.qg_env_solution.restype = i64
    L.qg_env_twists.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.qg_env_twists.restype = i64
    _lib = L
    return L


QG_DT_I8, QG_DT_F32, QG_DT_BF16, QG_DT_F16 = 0, 1, 2, 3


def check(status: int):
    if status != QG_OK:
        raise QGymError(status, load().qg_last_error().decode(errors="replace"))


def make_config(env_kind: str, num_qubits: int, *, metrics_weights=None, **fields) -> QGConfig:
    """Reference constructor defaults (clifford.rs:401-426, pauli.rs:743-778) overlaid with `fields`."""
    L = load()
    cfg = QGConfig()
    L.qg_config_default(C.byref(cfg), ENV_KIND[env_kind], int(num_qubits))
    for key, val in (metrics_weights or {}).items():  # metrics.rs:168-184: unknown keys ignored
        if key in ("n_cnots", "n_layers_cnots", "n_layers", "n_gates"):
            setattr(cfg, "w_" + key, float(val))
    for key, val in fields.items():
        if val is None:
            continue
        if not hasattr(cfg, key):
            raise TypeError(f"unknown config field {key!r}")
        setattr(cfg, key, val)
    return cfg


def make_gates(parsed):
    """`parsed`: list of (kind, q0, q1) from envs.gateset.parse_gateset."""
    arr = (QGGate * max(1, len(parsed)))()
    for i, (k, a, b) in enumerate(parsed):
        arr[i].kind, arr[i].q0, arr[i].q1 = k, a, b
    return arr
