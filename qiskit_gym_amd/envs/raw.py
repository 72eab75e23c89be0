"""RawEnv -- the scalar raw-env object the reference's Gym adapter wraps.

Mirror of `qiskit_gym_rs.{Clifford,LinearFunction,Permutation,PauliNetwork}Env` (PyO3 classes over
twisterl's `PyBaseEnv`; reference rust/src/envs/clifford.rs:384-427 and the method set listed at
src/qiskit_gym/envs/adapters.py:18-34), bound to the `qg_env_*` flavour of the C ABI: a batch of
one on the same HIP kernels.  Every call synchronises; use `qiskit_gym_amd.VecEnv` for throughput.

The reference draws its randomness from an unseedable RNG; here `reset()` takes a seed and `step()`
optionally the inversion coin, so runs are reproducible.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import _lib
from .gateset import parse_gateset


_MASK64 = 2**64 - 1
_PROCESS_ENTROPY = int.from_bytes(os.urandom(8), "little")  # differs per process, like thread_rng's OS seeding
_INSTANCE_COUNTER = itertools.count(1)


def _fresh_seed() -> int:
    """A seed no other env object of any process shares: per-process OS entropy mixed with a process-wide counter."""
    return (_PROCESS_ENTROPY ^ (0x9E3779B97F4A7C15 * next(_INSTANCE_COUNTER)) ^ int.from_bytes(os.urandom(8), "little")) & _MASK64


class RawEnv:
    """`seed=None` (default): the env draws like the reference's `thread_rng` (clifford.rs:266,307; pauli.rs:657) -- every
    object, and every clone, has its own unpredictable stream.  `seed=int`: reproducible streams (tests, parity runs)."""

    def __init__(self, env_kind: str, num_qubits: int, gateset: Sequence, device: int = 0,
                 metrics_weights: Optional[dict] = None, seed: Optional[int] = None, _handle=None, **config):
        self._L = _lib.load()
        self.env_kind = env_kind
        self.num_qubits = int(num_qubits)
        self.gateset = [(g[0], tuple(int(q) for q in g[1])) for g in gateset]
        self._reset_counter = 0
        self._seed = _fresh_seed() if seed is None else int(seed) & _MASK64
        if _handle is not None:
            self._h = _handle
            _lib.check(self._L.qg_env_set_seed(self._h, self._seed))
            return
        cfg = _lib.make_config(env_kind, num_qubits, metrics_weights=metrics_weights,
                               **{k: (int(v) if isinstance(v, bool) else v) for k, v in config.items()})
        gates = _lib.make_gates(parse_gateset(self.gateset))
        h = C.c_void_p()
        _lib.check(self._L.qg_env_create(C.byref(cfg), gates, len(self.gateset), int(device), C.byref(h)))
        self._h = h
        _lib.check(self._L.qg_env_set_seed(self._h, self._seed))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.qg_env_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def release_pool():
        """Free the handles that destroyed envs left in the process-wide clone pool (`qg_env_pool_clear`)."""
        _lib.load().qg_env_pool_clear()

    def clone(self, seed: Optional[int] = None) -> "RawEnv":
        """Deep copy of the state (Env: DynClone).  The clone's FUTURE random draws (reset scramble, add_inverts coins,
        PauliEnv permutations) are its own, as with the reference's thread_rng, unless `seed` pins them."""
        h = C.c_void_p()
        _lib.check(self._L.qg_env_clone(self._h, C.byref(h)))
        return RawEnv(self.env_kind, self.num_qubits, self.gateset, seed=seed, _handle=h)

    # ---- the Env trait (clifford.rs:285-382) --------------------------------------------------
    def num_actions(self) -> int:
        return int(self._L.qg_env_num_actions(self._h))

    def obs_shape(self) -> List[int]:
        out = (C.c_int64 * 2)()
        _lib.check(self._L.qg_env_obs_shape(self._h, out))
        return [int(out[0]), int(out[1])]

    @property
    def difficulty(self) -> int:
        return int(self._L.qg_env_get_difficulty(self._h))

    @difficulty.setter
    def difficulty(self, d: int):
        _lib.check(self._L.qg_env_set_difficulty(self._h, int(d)))

    def set_state(self, state: Sequence[int]):
        a = np.ascontiguousarray(np.asarray(list(state), dtype=np.int64))
        _lib.check(self._L.qg_env_set_state(self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), a.size))

    def reset(self, seed: Optional[int] = None):
        if seed is None:  # successive resets of one env, and resets of different envs / clones, draw different scrambles
            self._reset_counter += 1
            seed = self._seed ^ (0x9E3779B97F4A7C15 * self._reset_counter)
        _lib.check(self._L.qg_env_reset(self._h, int(seed) & (2**64 - 1)))

    def step(self, action: int, coin: Optional[int] = None):
        if coin is None:
            _lib.check(self._L.qg_env_step(self._h, int(action)))
        else:
            _lib.check(self._L.qg_env_step_coin(self._h, int(action), int(coin)))

    def masks(self) -> List[bool]:
        n = self.num_actions()
        buf = (C.c_uint8 * max(n, 1))()
        self._L.qg_env_masks(self._h, buf, n)
        return [bool(buf[i]) for i in range(n)]

    def is_final(self) -> bool:
        return bool(self._L.qg_env_is_final(self._h))

    def reward(self) -> float:
        return float(self._L.qg_env_reward(self._h))

    def success(self) -> bool:
        return bool(self._L.qg_env_success(self._h))

    def observe(self) -> List[int]:
        r, c = self.obs_shape()
        buf = (C.c_int64 * max(r * c, 1))()
        n = self._L.qg_env_observe(self._h, buf, r * c)
        if n < 0:
            _lib.check(int(n))
        return [int(buf[i]) for i in range(int(n))]

    def track_solution(self) -> bool:
        return bool(self._L.qg_env_track_solution(self._h))

    def solution(self) -> List[int]:
        n = self._L.qg_env_solution(self._h, None, 0)
        if n < 0:
            _lib.check(int(n))
        buf = (C.c_uint64 * max(int(n), 1))()
        self._L.qg_env_solution(self._h, buf, int(n))
        return [int(buf[i]) for i in range(int(n))]

    def twists(self) -> Tuple[List[List[int]], List[List[int]]]:
        n = self._L.qg_env_twists(self._h, None, None)
        if n < 0:
            _lib.check(int(n))
        n = int(n)
        if n == 0:
            return [], []
        r, c = self.obs_shape()
        osz, asz = r * c, self.num_actions()
        ob = (C.c_int64 * (n * osz))()
        ab = (C.c_int64 * (n * asz))()
        self._L.qg_env_twists(self._h, ob, ab)
        return ([[int(ob[i * osz + j]) for j in range(osz)] for i in range(n)],
                [[int(ab[i * asz + j]) for j in range(asz)] for i in range(n)])
