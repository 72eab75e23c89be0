"""GPU-resident rollout collection: policy forward, masked sampling, env.step and auto-reset all
stay on the device.

The reference collects episodes inside twisterl: rayon workers clone the scalar env per episode and
run a Rust copy of the policy on the CPU (`rl/synthesis.py:128-138`, notebook timing keys `collect`,
`data_to_torch`).  With the env batch resident in HBM the same loop is a handful of stream-ordered
launches per step and the trajectories are born as torch tensors -- there is no `data_to_torch`.

`BasicPolicy` mirrors the shape of the reference's default policy network (`twisterl.nn.BasicPolicy`
as configured by `BasicPolicyConfig`, `rl/configs.py:531-607`; checkpoint shapes in
`examples/models/*.pt`): Linear(prod(obs_shape) -> 512) -> ReLU -> Linear(512 -> 256) -> ReLU ->
{Linear(256 -> num_actions), Linear(256 -> 1)}.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .vec import VecEnv


class BasicPolicy(nn.Module):
    def __init__(self, obs_size: int, num_actions: int, embedding_size: int = 512, common: int = 256):
        super().__init__()
        self.embeddings = nn.Linear(obs_size, embedding_size)
        self.common = nn.Linear(embedding_size, common)
        self.policy_head = nn.Linear(common, num_actions)
        self.value_head = nn.Linear(common, 1)

    def forward(self, obs_flat: torch.Tensor):
        h = torch.relu(self.embeddings(obs_flat))
        h = torch.relu(self.common(h))
        return self.policy_head(h), self.value_head(h).squeeze(-1)


@dataclass
class Rollout:
    obs: torch.Tensor       # [T, B, rows*cols] int8 (dense observation before each action)
    actions: torch.Tensor   # [T, B] int64
    logp: torch.Tensor      # [T, B] float32
    values: torch.Tensor    # [T, B] float32
    rewards: torch.Tensor   # [T, B] float32
    dones: torch.Tensor     # [T, B] uint8 (episode ended with this step)


class RolloutCollector:
    """Steps `env` for T steps under `policy`, resetting finished episodes on the device."""

    def __init__(self, env: VecEnv, policy: nn.Module, dtype: torch.dtype = torch.bfloat16, seed: int = 0):
        self.env = env
        self.policy = policy.to(device=env.device, dtype=dtype)
        self.dtype = dtype
        self.seed = int(seed)
        self.steps_done = 0
        r, c = env.obs_shape_
        self.obs_size = r * c
        self._gen = torch.Generator(device=env.device)
        self._gen.manual_seed(self.seed)

    @torch.no_grad()
    def collect(self, T: int, out: Optional[Rollout] = None) -> Rollout:
        env, B, dev = self.env, self.env.batch, self.env.device
        if out is None:
            out = Rollout(
                obs=torch.empty((T, B, self.obs_size), dtype=torch.int8, device=dev),
                actions=torch.empty((T, B), dtype=torch.int64, device=dev),
                logp=torch.empty((T, B), dtype=torch.float32, device=dev),
                values=torch.empty((T, B), dtype=torch.float32, device=dev),
                rewards=torch.empty((T, B), dtype=torch.float32, device=dev),
                dones=torch.empty((T, B), dtype=torch.uint8, device=dev),
            )
        for t in range(T):
            # finished episodes start over (reference: the collector calls reset() on a fresh clone)
            env.reset_done(self.seed + 0x9E3779B9 * (self.steps_done + 1))
            env.observe(out=out.obs[t].view(B, *env.obs_shape_))
            logits, value = self.policy(out.obs[t].to(self.dtype))
            logp_all = torch.log_softmax(logits.float(), dim=-1)
            # masks() is all-true for a live env (clifford.rs:349-351), so sampling needs no masking
            act = torch.multinomial(logp_all.exp(), 1, generator=self._gen).squeeze(1)
            out.actions[t] = act
            out.logp[t] = logp_all.gather(1, act.unsqueeze(1)).squeeze(1)
            out.values[t] = value.float()
            env.step(act)
            out.rewards[t].copy_(env.reward)
            out.dones[t].copy_(env.done)
            self.steps_done += 1
        return out
