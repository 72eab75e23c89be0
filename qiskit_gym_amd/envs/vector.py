"""VecGym -- a `gymnasium.vector.VectorEnv`-shaped front end over the GPU-resident batch.

The reference has no vector env: twisterl clones scalar envs across rayon workers, and Gymnasium users wrap
`gym_adapter` envs in `gymnasium.vector.SyncVectorEnv` (one Python env per copy).  This class gives that calling
convention to one `VecEnv` handle: `reset() -> (obs, {})`, `step(actions) -> (obs, reward, terminated, truncated, {})`
with tensors of leading dimension `num_envs`, and SAME-STEP autoreset -- a finished env is re-scrambled on the device
(`qg_vec_reset_done`) inside the step that finished it, so the returned observation is already the next episode's
first one while `reward / terminated / truncated` describe the step just taken.  `terminated` = solved
(`Env::success`), `truncated` = out of depth without being solved (`is_final` and not `success`, clifford.rs:353).

Randomness: episode n of the handle's lifetime is scrambled with seed `seed + 0x9E3779B9 * (n + 1)` (the collector's
rule), so a run is reproducible and can be replayed on the CPU oracle (tests/test_gpu_gyms.py)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ..vec import VecEnv
from .frontend import _Discrete, _MultiBinary


class VecGym:
    def __init__(self, venv: VecEnv, seed: int = 0):
        self.venv = venv
        self.num_envs = venv.batch
        self.single_observation_space = _MultiBinary(tuple(venv.obs_shape_))
        self.single_action_space = _Discrete(venv.num_actions())
        self.seed = int(seed)
        self._episodes = 0  # reset rounds so far

    def _next_seed(self) -> int:
        self._episodes += 1
        return self.seed + 0x9E3779B9 * self._episodes

    def reset(self, *, seed: Optional[int] = None, options=None) -> Tuple[torch.Tensor, dict]:
        if seed is not None:
            self.seed, self._episodes = int(seed), 0
        self.venv.reset(self._next_seed())
        return self.venv.observe(), {}

    def step(self, actions: torch.Tensor):
        v = self.venv
        v.step(actions)
        reward = v.reward.clone()
        terminated = v.success.bool()
        truncated = v.done.bool() & ~terminated
        v.reset_done(self._next_seed())  # only the envs whose episode just ended; stream-ordered, no host round trip
        return v.observe(), reward, terminated, truncated, {}

    def close(self):
        self.venv.close()
