"""Multi-GPU sharding of a batch of environments (one process per GPU, RCCL over xGMI).

Environments are independent, so `step` needs no collective: rank r of W owns the contiguous env
range `shard_range(total, r, W)` and steps it with its own `VecEnv`.  The one exchange on the path
is the observation handed to the learner: ranks all-gather the *bit-packed* observation
(CliffordGym 16q: 128 B/env instead of 1 KiB/env dense int8) and unpack locally.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """(first env, number of envs) owned by `rank`: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total_envs), int(world))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def local_actions(global_actions: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Slice of a `[..., total_envs]` action tensor that belongs to `rank`."""
    start, count = shard_range(global_actions.shape[-1], rank, world)
    return global_actions[..., start:start + count]


def all_gather_observation(local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """All-gather equal-sized shards `[B_local, ...]` into `[W * B_local, ...]` in rank order
    (`all_gather_into_tensor`: RCCL on GPUs, gloo on CPU tensors)."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo":  # gloo has no all_gather_into_tensor
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, local.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def unpack_rows_u32(packed: torch.Tensor, dim: int) -> torch.Tensor:
    """Bit-packed rows `[B, D]` (int32 words, bit c = entry (r, c)) -> dense int8 `[B, D, dim]`.
    Reference-layout helper for consumers of the gathered tensor (pure torch, any device)."""
    shifts = torch.arange(dim, device=packed.device, dtype=torch.int64)
    words = packed.to(torch.int64) & 0xFFFFFFFF
    return ((words.unsqueeze(-1) >> shifts) & 1).to(torch.int8)
