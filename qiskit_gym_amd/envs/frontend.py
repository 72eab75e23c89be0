"""GymFrontEnd -- the Gymnasium-facing shell of one scalar environment.

Behavioural contract (what the reference's `gym_adapter` wrapper gives its users,
src/qiskit_gym/envs/adapters.py:36-105): `observation_space = MultiBinary(obs_shape)`,
`action_space = Discrete(num_actions)`, `reset() -> (obs, {})`, `step(a) -> (obs, reward,
terminated, False, {})` with `terminated = is_final()` and an assertion that the env is not final
before the step, dense int8 observations rebuilt from the sparse `observe()` indices, unknown
attributes forwarded to the raw env, `env.difficulty = d` forwarded, `to_json()` = the constructor
kwargs.  Here it is a plain base class over `RawEnv` (a batch of one on the GPU kernels) instead
of a class-generating decorator.  gymnasium is optional: present, the class derives from
`gymnasium.Env` and uses its spaces; absent (as in the build image), equivalent stand-ins are used.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np

from .raw import RawEnv

try:  # pragma: no cover - gymnasium is not installed in the build image
    import gymnasium as _gym

    _EnvBase = _gym.Env
    _MultiBinary, _Discrete = _gym.spaces.MultiBinary, _gym.spaces.Discrete
except Exception:
    class _EnvBase:  # the slice of gymnasium.Env the front end relies on
        metadata: Dict[str, Any] = {}

        def reset(self, *, seed: Optional[int] = None, options=None):
            return None

    class _MultiBinary:
        def __init__(self, shape):
            self.n = tuple(shape)
            self.shape = tuple(shape)
            self.dtype = np.dtype(np.int8)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.isin(x, (0, 1)).all())

        def __repr__(self):
            return f"MultiBinary({self.n})"

    class _Discrete:
        def __init__(self, n: int):
            self.n = int(n)

        def contains(self, x) -> bool:
            return 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"


class GymFrontEnd(_EnvBase):
    """Subclasses set `env_kind`; the constructor kwargs are the reference's (num_qubits, gateset, ...)."""

    metadata = {"render_modes": ["human"], "render_fps": 4}
    env_kind: str = ""
    _FORWARDED_SETTERS = ("difficulty",)

    def __init__(self, **config):
        object.__setattr__(self, "config", dict(config))
        raw_cfg = dict(config)
        raw = RawEnv(self.env_kind, raw_cfg.pop("num_qubits"), raw_cfg.pop("gateset"), **raw_cfg)
        object.__setattr__(self, "_raw_env", raw)
        shape = tuple(raw.obs_shape())
        object.__setattr__(self, "_obs_shape", shape)
        object.__setattr__(self, "_obs_size", int(np.prod(shape)))
        self.observation_space = _MultiBinary(shape)
        self.action_space = _Discrete(raw.num_actions())

    # -- observation -----------------------------------------------------------------------------
    def _dense_observation(self) -> np.ndarray:
        dense = np.zeros(self._obs_size, dtype=np.int8)
        dense[np.asarray(self._raw_env.observe(), dtype=np.int64)] = 1
        return dense.reshape(self._obs_shape)

    # -- gymnasium API ---------------------------------------------------------------------------
    def reset(self, *, seed: Optional[int] = None, options=None) -> Tuple[np.ndarray, dict]:
        super().reset(seed=seed)
        self._raw_env.reset(seed)  # the reference's Rust side ignores the seed; here it is honoured
        return self._dense_observation(), {}

    def step(self, action):
        if self._raw_env.is_final():
            raise AssertionError("Action provided when env is in final state.")
        self._raw_env.step(int(action))
        return self._dense_observation(), float(self._raw_env.reward()), bool(self._raw_env.is_final()), False, {}

    def render(self, mode: str = "human"):
        print(self._dense_observation())

    def close(self):
        self._raw_env.close() if hasattr(self._raw_env, "close") else None

    def to_json(self) -> dict:
        return self.config

    # -- attribute plumbing ------------------------------------------------------------------------
    def __getattr__(self, name):  # only reached for names the front end itself does not define
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(object.__getattribute__(self, "_raw_env"), name)

    def __setattr__(self, name, value):
        if name in self._FORWARDED_SETTERS and "_raw_env" in self.__dict__:
            setattr(self._raw_env, name, value)
        else:
            object.__setattr__(self, name, value)


def gym_adapter(env_kind: str):
    """Compatibility helper: a GymFrontEnd subclass bound to `env_kind`."""
    return type(f"{env_kind.title().replace('_', '')}EnvGym", (GymFrontEnd,), {"env_kind": env_kind})
