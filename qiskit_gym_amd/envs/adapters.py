"""gym_adapter -- Gymnasium-style wrapper around a raw env.

Same behaviour as the reference's decorator (src/qiskit_gym/envs/adapters.py:18-105): dense int8
observation built from the sparse `observe()` indices, the 5-tuple `step`, `reset -> (obs, {})`,
attribute forwarding, `difficulty` assignment forwarding and `to_json`.  gymnasium is optional in
this environment: when it is importable the wrapper subclasses `gymnasium.Env` and uses its
spaces, otherwise small stand-ins with the same attributes are used.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is not installed in the build image
    import gymnasium as gym
    from gymnasium import spaces

    _Base = gym.Env
    MultiBinary, Discrete = spaces.MultiBinary, spaces.Discrete
except Exception:  # gymnasium absent: minimal stand-ins

    class _Base:  # noqa: D401 - same surface as gymnasium.Env for what the adapter uses
        def reset(self, *, seed=None, options=None):
            return None

    class MultiBinary:
        def __init__(self, n):
            self.n = tuple(n) if hasattr(n, "__len__") else n
            self.shape = tuple(n) if hasattr(n, "__len__") else (n,)
            self.dtype = np.int8

        def __repr__(self):
            return f"MultiBinary({self.n})"

    class Discrete:
        def __init__(self, n):
            self.n = int(n)

        def __repr__(self):
            return f"Discrete({self.n})"


def gym_adapter(cls):
    class GymWrapper(_Base):
        metadata = {"render_modes": ["human"], "render_fps": 4}

        def __init__(self, *args, **kwargs):
            self.config = kwargs.copy()
            self._raw_env = cls(*args, **kwargs)
            self._obs_shape = tuple(self._raw_env.obs_shape())
            self.observation_space = MultiBinary(self._obs_shape)
            self.action_space = Discrete(self._raw_env.num_actions())

        def _full_obs(self):
            full = np.zeros(int(np.prod(self._obs_shape)), dtype=np.int8)
            full[self._raw_env.observe()] = 1
            return full.reshape(self._obs_shape)

        def reset(self, *, seed=None, options=None):
            super().reset(seed=seed)
            # the reference ignores `seed` on the Rust side; here it makes the scramble reproducible
            self._raw_env.reset(seed)
            return self._full_obs(), {}

        def step(self, action):
            assert not bool(self._raw_env.is_final()), "Action provided when env is in final state."
            self._raw_env.step(int(action))
            obs = self._full_obs()
            reward = float(self._raw_env.reward())
            terminated = bool(self._raw_env.is_final())
            return obs, reward, terminated, False, {}

        def render(self, mode="human"):
            print(self._full_obs())

        def close(self):
            if hasattr(self._raw_env, "close"):
                self._raw_env.close()

        def __getattr__(self, name):
            if name == "_raw_env":
                raise AttributeError(name)
            return getattr(self._raw_env, name)

        def __setattr__(self, name, value):
            if name in ("difficulty",) and "_raw_env" in self.__dict__:
                setattr(self._raw_env, name, value)
            else:
                super().__setattr__(name, value)

        def to_json(self):
            return self.config

    GymWrapper.__name__ = f"{cls.__name__}Gym"
    return GymWrapper
