"""ctypes front-end for the CPU oracle (oracle/libqgym_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing in qiskit_gym_amd/ imports this module.

The wrapper mirrors the reference's raw-env API (the method set `gym_adapter` expects,
/root/reference/src/qiskit_gym/envs/adapters.py:18-34) so parity tests read like reference tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
# QGYM_ORACLE_ASAN=1 (with LD_PRELOAD=$(gcc -print-file-name=libasan.so)): run the CPU tests on the address/UB-sanitizer build
_LIB_NAME = "libqgym_oracle_asan.so" if os.environ.get("QGYM_ORACLE_ASAN") else "libqgym_oracle.so"
_LIB_PATH = os.path.join(ORACLE_DIR, _LIB_NAME)

GATE_KINDS = {"h": 0, "s": 1, "sdg": 2, "sx": 3, "sxdg": 4, "cx": 5, "cz": 6, "swap": 7}
ENV_KINDS = {"permutation": 0, "linear_function": 1, "clifford": 2, "pauli": 3}


class OGGate(C.Structure):
    _fields_ = [("kind", C.c_int32), ("q0", C.c_int32), ("q1", C.c_int32)]


class OGConfig(C.Structure):
    _fields_ = [
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


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the .so is missing or stale."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("qgym_oracle.c", "qgym_oracle_symmetry.c", "qgym_oracle.h", "Makefile")]
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    )
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR, _LIB_NAME], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, sz, i64p, u8p, i32p, f32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.og_config_default.argtypes = [C.POINTER(OGConfig), C.c_int32, C.c_int32]
    L.og_env_new.restype = vp
    L.og_env_new.argtypes = [C.c_int32, C.POINTER(OGConfig), C.POINTER(OGGate), sz]
    L.og_env_clone.restype = vp
    L.og_env_clone.argtypes = [vp]
    L.og_env_free.argtypes = [vp]
    L.og_last_error.restype = C.c_char_p
    L.og_env_num_actions.restype = sz
    L.og_env_num_actions.argtypes = [vp]
    L.og_env_obs_shape.restype = sz
    L.og_env_obs_shape.argtypes = [vp, C.POINTER(sz)]
    L.og_env_set_difficulty.argtypes = [vp, sz]
    L.og_env_get_difficulty.restype = sz
    L.og_env_get_difficulty.argtypes = [vp]
    L.og_env_set_state.argtypes = [vp, i64p, sz]
    L.og_env_reset_with.argtypes = [vp, i64p, sz]
    L.og_env_step.argtypes = [vp, sz, C.c_int]
    L.og_env_masks.restype = sz
    L.og_env_masks.argtypes = [vp, u8p, sz]
    L.og_env_is_final.argtypes = [vp]
    L.og_env_reward.restype = C.c_float
    L.og_env_reward.argtypes = [vp]
    L.og_env_success.argtypes = [vp]
    L.og_env_observe.restype = sz
    L.og_env_observe.argtypes = [vp, i64p, sz, sz]
    L.og_env_track_solution.argtypes = [vp]
    L.og_env_solution.restype = sz
    L.og_env_solution.argtypes = [vp, C.POINTER(C.c_uint64), sz]
    L.og_env_depth.restype = sz
    L.og_env_depth.argtypes = [vp]
    L.og_env_inverted.argtypes = [vp]
    L.og_env_get_state_i64.restype = sz
    L.og_env_get_state_i64.argtypes = [vp, i64p, sz]
    L.og_env_metrics.argtypes = [vp, C.POINTER(sz)]
    L.og_pauli_reset_from.argtypes = [vp, u8p, C.c_char_p, sz]
    L.og_pauli_reset_seeded.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.og_pauli_set_perms.argtypes = [vp, i64p, i64p, sz]
    L.og_pauli_active_rotations.restype = sz
    L.og_pauli_active_rotations.argtypes = [vp, i64p, sz]
    L.og_pauli_rotation.argtypes = [vp, sz, u8p, u8p, i32p, i32p]
    L.og_pauli_num_rotations.restype = sz
    L.og_pauli_num_rotations.argtypes = [vp]
    L.og_vec_new.restype = vp
    L.og_vec_new.argtypes = [vp, sz]
    L.og_vec_free.argtypes = [vp]
    L.og_env_twists.restype = C.c_int64
    L.og_env_twists.argtypes = [vp, i64p, i64p]
    L.og_twists.restype = C.c_int64
    L.og_twists.argtypes = [C.c_int32, sz, C.POINTER(OGGate), sz, i64p, i64p]
    L.og_qubit_perms.restype = C.c_int64
    L.og_qubit_perms.argtypes = [sz, C.POINTER(OGGate), sz, i64p, i64p]
    L.og_vec_env.restype = vp
    L.og_vec_env.argtypes = [vp, sz]
    L.og_vec_set_state.argtypes = [vp, i64p, sz]
    L.og_vec_reset_with.argtypes = [vp, i64p, sz]
    L.og_vec_step.argtypes = [vp, i32p, u8p, f32p, u8p, u8p, i32p, C.c_int]
    L.og_vec_observe_dense.argtypes = [vp, C.POINTER(C.c_int8), C.c_int]
    L.og_vec_get_state.argtypes = [vp, i64p, sz]
    u64p = C.POINTER(C.c_uint64)
    L.og_vec_reset_seeded.argtypes = [vp, C.c_uint64, C.c_uint64, u64p, u8p, C.c_int]
    L.og_vec_pauli_reset_seeded.argtypes = [vp, C.c_uint64, C.c_uint64, u64p, u8p, C.c_int]
    L.og_vec_solutions.argtypes = [vp, u64p, sz, i64p]
    _lib = L
    return L


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def make_gates(gateset: Sequence[Tuple[str, Sequence[int]]]):
    arr = (OGGate * max(1, len(gateset)))()
    for i, (name, idx) in enumerate(gateset):
        k = GATE_KINDS[name.strip().lower()]
        arr[i].kind = k
        arr[i].q0 = int(idx[0])
        arr[i].q1 = int(idx[1]) if len(idx) > 1 else 0
    return arr


class OracleError(RuntimeError):
    """Raised where the reference would panic."""


class OracleEnv:
    """One scalar reference env (Clifford / LinearFunction / Permutation / PauliNetwork)."""

    def __init__(self, kind: str, num_qubits: int, gateset, *, _handle=None, **kw):
        L = lib()
        self.kind = kind
        self.num_qubits = num_qubits
        self.gateset = [(n, tuple(int(q) for q in idx)) for n, idx in gateset]
        if _handle is not None:
            self._h = _handle
            return
        cfg = OGConfig()
        L.og_config_default(C.byref(cfg), ENV_KINDS[kind], num_qubits)
        weights = kw.pop("metrics_weights", None) or {}
        for key, val in weights.items():  # metrics.rs:168-184: unknown keys ignored
            if key in ("n_cnots", "n_layers_cnots", "n_layers", "n_gates"):
                setattr(cfg, "w_" + key, float(val))
        if kw.get("final_pauli_layers", 0) is None:
            kw.pop("final_pauli_layers")
        for key, val in kw.items():
            if not hasattr(cfg, key):
                raise TypeError(f"unknown config field {key}")
            setattr(cfg, key, val)
        self._gates = make_gates(self.gateset)
        self._h = L.og_env_new(ENV_KINDS[kind], C.byref(cfg), self._gates, len(self.gateset))
        if not self._h:
            raise OracleError(L.og_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_owned", True):
            lib().og_env_free(self._h)
            self._h = None

    def _check(self, rc):
        if rc != 0:
            raise OracleError(lib().og_last_error().decode())

    def clone(self) -> "OracleEnv":
        return OracleEnv(self.kind, self.num_qubits, self.gateset, _handle=lib().og_env_clone(self._h))

    # --- Env trait ---
    def num_actions(self) -> int:
        return lib().og_env_num_actions(self._h)

    def obs_shape(self) -> List[int]:
        out = (C.c_size_t * 2)()
        lib().og_env_obs_shape(self._h, out)
        return [out[0], out[1]]

    @property
    def difficulty(self) -> int:
        return lib().og_env_get_difficulty(self._h)

    @difficulty.setter
    def difficulty(self, d: int):
        lib().og_env_set_difficulty(self._h, int(d))

    def set_state(self, state: Iterable[int]):
        a = np.ascontiguousarray(np.asarray(list(state), dtype=np.int64))
        self._check(lib().og_env_set_state(self._h, _ptr(a, C.c_int64), a.size))

    def reset_with(self, actions: Iterable[int]):
        a = np.ascontiguousarray(np.asarray(list(actions), dtype=np.int64))
        self._check(lib().og_env_reset_with(self._h, _ptr(a, C.c_int64), a.size))

    def step(self, action: int, coin: int = 0):
        a = int(action)
        self._check(lib().og_env_step(self._h, a if a >= 0 else (1 << 64) - 1, int(coin)))

    def masks(self) -> List[bool]:
        n = self.num_actions()
        buf = np.zeros(max(n, 1), dtype=np.uint8)
        lib().og_env_masks(self._h, _ptr(buf, C.c_uint8), n)
        return [bool(x) for x in buf[:n]]

    def is_final(self) -> bool:
        return bool(lib().og_env_is_final(self._h))

    def reward(self) -> float:
        return float(lib().og_env_reward(self._h))

    def reward_bits(self) -> int:
        return int(np.float32(lib().og_env_reward(self._h)).view(np.uint32))

    def success(self) -> bool:
        return bool(lib().og_env_success(self._h))

    def observe(self, perm_idx: int = 0) -> List[int]:
        shp = self.obs_shape()
        cap = shp[0] * shp[1]
        buf = np.zeros(max(cap, 1), dtype=np.int64)
        n = lib().og_env_observe(self._h, _ptr(buf, C.c_int64), cap, perm_idx)
        return buf[:n].tolist()

    def dense_obs(self, perm_idx: int = 0) -> np.ndarray:
        shp = self.obs_shape()
        full = np.zeros(shp[0] * shp[1], dtype=np.int8)  # adapters.py:50-54
        full[self.observe(perm_idx)] = 1
        return full.reshape(shp)

    def track_solution(self) -> bool:
        return bool(lib().og_env_track_solution(self._h))

    def solution(self) -> List[int]:
        n = lib().og_env_solution(self._h, None, 0)
        buf = np.zeros(max(n, 1), dtype=np.uint64)
        lib().og_env_solution(self._h, _ptr(buf, C.c_uint64), n)
        return [int(x) for x in buf[:n]]

    # --- white box ---
    def twists(self) -> Tuple[List[List[int]], List[List[int]]]:
        """Env::twists (clifford.rs:370-372): (obs_perms, act_perms)."""
        n = int(lib().og_env_twists(self._h, None, None))
        if n == 0:
            return [], []
        r, c = self.obs_shape()
        ob = np.zeros((n, r * c), dtype=np.int64)
        ab = np.zeros((n, max(self.num_actions(), 1)), dtype=np.int64)
        lib().og_env_twists(self._h, _ptr(ob, C.c_int64), _ptr(ab, C.c_int64))
        return ob.tolist(), ab[:, : self.num_actions()].tolist()

    def depth(self) -> int:
        return lib().og_env_depth(self._h)

    def inverted(self) -> bool:
        return bool(lib().og_env_inverted(self._h))

    def get_state(self) -> np.ndarray:
        if self.kind == "permutation":
            cap = self.num_qubits
        elif self.kind == "linear_function":
            cap = self.num_qubits**2
        else:
            cap = 4 * self.num_qubits**2
        buf = np.zeros(max(cap, 1), dtype=np.int64)
        n = lib().og_env_get_state_i64(self._h, _ptr(buf, C.c_int64), cap)
        return buf[:n]

    def metrics(self) -> Tuple[int, int, int, int]:
        out = (C.c_size_t * 4)()
        lib().og_env_metrics(self._h, out)
        return tuple(out)

    # --- Pauli ---
    def pauli_reset_from(self, tableau: np.ndarray, labels: Sequence[str]):
        t = np.ascontiguousarray(np.asarray(tableau, dtype=np.uint8).reshape(-1))
        blob = b"".join(s.encode() + b"\0" for s in labels) + b"\0"
        self._check(lib().og_pauli_reset_from(self._h, _ptr(t, C.c_uint8), blob, len(labels)))

    def pauli_reset_seeded(self, seed: int, env_index: int):
        self._check(lib().og_pauli_reset_seeded(self._h, int(seed), int(env_index)))

    def pauli_set_perms(self, qubit_perms, act_perms):
        qp = np.ascontiguousarray(np.asarray(qubit_perms, dtype=np.int64).reshape(-1))
        ap = np.ascontiguousarray(np.asarray(act_perms, dtype=np.int64).reshape(-1))
        self._check(lib().og_pauli_set_perms(self._h, _ptr(qp, C.c_int64), _ptr(ap, C.c_int64), len(qubit_perms)))

    def active_rotations(self) -> List[int]:
        buf = np.zeros(64, dtype=np.int64)
        n = lib().og_pauli_active_rotations(self._h, _ptr(buf, C.c_int64), 64)
        return buf[:n].tolist()

    def num_rotations(self) -> int:
        return lib().og_pauli_num_rotations(self._h)

    def rotation(self, k: int):
        x = np.zeros(self.num_qubits, dtype=np.uint8)
        z = np.zeros(self.num_qubits, dtype=np.uint8)
        bp, ph = C.c_int32(), C.c_int32()
        self._check(lib().og_pauli_rotation(self._h, k, _ptr(x, C.c_uint8), _ptr(z, C.c_uint8), C.byref(bp), C.byref(ph)))
        return x, z, bp.value, ph.value


class OracleVec:
    """B clones of one prototype env stepped side by side on the CPU."""

    def __init__(self, proto: OracleEnv, batch: int):
        self.proto = proto
        self.batch = batch
        self._h = lib().og_vec_new(proto._h, batch)
        shp = proto.obs_shape()
        self.obs_size = shp[0] * shp[1]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().og_vec_free(self._h)
            self._h = None

    def env(self, i: int) -> OracleEnv:
        e = OracleEnv(self.proto.kind, self.proto.num_qubits, self.proto.gateset, _handle=lib().og_vec_env(self._h, i))
        e._owned = False
        return e

    def set_state(self, states: np.ndarray):
        s = np.ascontiguousarray(np.asarray(states, dtype=np.int64).reshape(self.batch, -1))
        if lib().og_vec_set_state(self._h, _ptr(s, C.c_int64), s.shape[1]) != 0:
            raise OracleError(lib().og_last_error().decode())

    def reset_with(self, actions: np.ndarray):
        """actions: [n_steps, B] scramble draws."""
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.int64).reshape(-1, self.batch))
        if lib().og_vec_reset_with(self._h, _ptr(a, C.c_int64), a.shape[0]) != 0:
            raise OracleError(lib().og_last_error().decode())

    def reset_seeded(self, seed: int, env_ids=None, env_base: int = 0, mask=None, threads: int = 0):
        """Env::reset of the envs selected by `mask` (None = all), its draws from the counter RNG of the HIP path keyed by the global env id
        (env_ids[i] or env_base + i): what qg_vec_reset(seed) / qg_vec_reset_done(seed) do on the device.  PauliEnv: the whole target generator."""
        ids = ip = None
        if env_ids is not None:
            ids = np.ascontiguousarray(np.asarray(env_ids, dtype=np.uint64).reshape(self.batch))
            ip = _ptr(ids, C.c_uint64)
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8).reshape(self.batch))
            mp = _ptr(mask, C.c_uint8)
        fn = lib().og_vec_pauli_reset_seeded if self.proto.kind == "pauli" else lib().og_vec_reset_seeded
        if fn(self._h, int(seed) & (2**64 - 1), int(env_base), ip, mp, threads) != 0:
            raise OracleError(lib().og_last_error().decode())

    def solutions(self, cap: int):
        """Env::solution of every env: (entries uint64 [B, cap], lengths int64 [B])."""
        out = np.zeros((self.batch, max(cap, 1)), dtype=np.uint64)
        lens = np.zeros(self.batch, dtype=np.int64)
        lib().og_vec_solutions(self._h, _ptr(out, C.c_uint64), cap, _ptr(lens, C.c_int64))
        return out[:, :cap], lens

    def step(self, actions: np.ndarray, coins: np.ndarray | None = None, threads: int = 0):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.int32).reshape(self.batch))
        reward = np.zeros(self.batch, dtype=np.float32)
        success = np.zeros(self.batch, dtype=np.uint8)
        final = np.zeros(self.batch, dtype=np.uint8)
        depth = np.zeros(self.batch, dtype=np.int32)
        cp = None
        if coins is not None:
            coins = np.ascontiguousarray(np.asarray(coins, dtype=np.uint8).reshape(self.batch))
            cp = _ptr(coins, C.c_uint8)
        rc = lib().og_vec_step(
            self._h, _ptr(a, C.c_int32), cp, _ptr(reward, C.c_float), _ptr(success, C.c_uint8), _ptr(final, C.c_uint8),
            _ptr(depth, C.c_int32), threads)
        if rc != 0:
            raise OracleError(lib().og_last_error().decode() or "og_vec_step failed")
        return reward, success, final, depth

    def step_only(self, actions: np.ndarray, threads: int = 0):
        """Timing variant: no per-env outputs."""
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.int32).reshape(self.batch))
        return lib().og_vec_step(self._h, _ptr(a, C.c_int32), None, None, None, None, None, threads)

    def observe_dense(self, threads: int = 0) -> np.ndarray:
        out = np.zeros((self.batch, self.obs_size), dtype=np.int8)
        lib().og_vec_observe_dense(self._h, _ptr(out, C.c_int8), threads)
        return out

    def get_state(self, per_env: int) -> np.ndarray:
        out = np.empty((self.batch, per_env), dtype=np.int64)
        if lib().og_vec_get_state(self._h, _ptr(out, C.c_int64), per_env) != 0:
            raise OracleError(lib().og_last_error().decode())
        return out


def qubit_perms(num_qubits: int, gateset) -> Tuple[List[List[int]], List[List[int]]]:
    """compute_qubit_perms (symmetry.rs:307-361): (qubit_perms, act_perms)."""
    gates = make_gates(gateset)
    n = int(lib().og_qubit_perms(num_qubits, gates, len(gateset), None, None))
    qp = np.zeros((max(n, 1), max(num_qubits, 1)), dtype=np.int64)
    ap = np.zeros((max(n, 1), max(len(gateset), 1)), dtype=np.int64)
    lib().og_qubit_perms(num_qubits, gates, len(gateset), _ptr(qp, C.c_int64), _ptr(ap, C.c_int64))
    return qp[:n, :num_qubits].tolist(), ap[:n, : len(gateset)].tolist()
