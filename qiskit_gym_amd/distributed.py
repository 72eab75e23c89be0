"""Multi-GPU sharding of a batch of environments (one process per GPU, RCCL over xGMI).

Environments are independent, so `step` needs no collective: rank r of W owns the contiguous env
range `shard_range(total, r, W)` and steps it with its own `VecEnv`.  The one exchange on the path
is the observation handed to the learner: ranks all-gather the *bit-packed* observation
(CliffordGym 16q: 128 B/env instead of 1 KiB/env dense int8) and unpack locally.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

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


def device_for_rank(local_rank: int, world: int, visible: int, share_gpu0: bool = False) -> int:
    """The GPU a rank uses: its LOCAL_RANK (one process per GPU, whatever the node shows beyond that), or GPU 0 for every rank of a functional
    rehearsal on a one-GPU box (`share_gpu0`).  Raises when that GPU is not visible -- never falls back to another one."""
    if visible < 1:
        raise RuntimeError("no GPU is visible")
    if share_gpu0:
        return 0
    if not 0 <= local_rank < visible:
        raise RuntimeError(f"LOCAL_RANK {local_rank} of a {world}-rank job, but only {visible} GPU(s) are visible")
    return local_rank


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


def all_ranks_ok(ok: bool, group=None) -> bool:
    """Collective vote over the control plane (CPU tensor: gloo): True when EVERY rank passed True."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def run_guarded_phases(phases, group=None) -> Optional[dict]:
    """Run an OPTIONAL multi-rank leg (e.g. a measurement beside the headline) so that one rank's failure cannot strand the others.

    `phases`: [(name, callable)].  A phase may fail on one rank only (mapping a peer's memory, a device-side deadline); after every
    phase all ranks vote (`all_ranks_ok`), and at the first phase that did not pass everywhere EVERY rank stops there -- no rank goes on
    into a barrier its peers never reach.  Returns None when all phases passed on all ranks, else
    `{"error", "phase", "failed_on_this_rank"}` (plus `"control_plane"` when the vote itself failed: a peer died or the control plane's
    timeout expired -- the group is then unusable and the caller should finish without further collectives)."""
    for name, fn in phases:
        err = None
        try:
            fn()
        except Exception as exc:  # noqa: BLE001 -- whatever went wrong on this rank, the peers must hear of it
            err = f"{type(exc).__name__}: {exc}"
        try:
            ok = all_ranks_ok(err is None, group)
        except Exception as exc:  # noqa: BLE001
            return {"error": err or "the vote after this phase failed", "phase": name, "failed_on_this_rank": err is not None,
                    "control_plane": f"{type(exc).__name__}: {exc}"}
        if not ok:
            return {"error": err or "a peer rank failed in this phase", "phase": name, "failed_on_this_rank": err is not None}
    return None


def learner_shard_words(batch: int, obs_words: int) -> int:
    """int32 words of one rank's hand-over shard: packed observation [batch, obs_words], f32 rewards [batch], `is_final` and
    `success` bytes [batch] each (SURVEY.md 8e's three gathers as one flat buffer, every section 4-byte aligned, the whole
    shard padded to 16 bytes) -- `qg_shard_layout` of include/qgym.h for 32-bit observation words."""
    return (batch * obs_words + batch + 2 * ((batch + 3) // 4) + 3) // 4 * 4


def fill_learner_shard(buf: torch.Tensor, batch: int, obs_words: int, write_obs: Callable[[torch.Tensor], None], reward: torch.Tensor,
                       done: torch.Tensor, success: torch.Tensor):
    """Fill one rank's flat int32 shard in place: `write_obs(view)` writes the packed observation into its [batch, obs_words] view."""
    pad = 4 * ((batch + 3) // 4)
    write_obs(buf[: batch * obs_words].view(batch, obs_words))
    buf[batch * obs_words : batch * (obs_words + 1)].copy_(reward.view(torch.int32))
    flags = buf[batch * (obs_words + 1) :].view(torch.uint8)
    flags[:batch].copy_(done)
    flags[pad : pad + batch].copy_(success)


def split_learner_shards(gathered: torch.Tensor, world: int, batch: int, obs_words: int):
    """The all-gathered `[world * learner_shard_words]` buffer -> (obs [world*batch, obs_words] int32, reward [world*batch] f32,
    done, success [world*batch] uint8), rank order = env order."""
    n = learner_shard_words(batch, obs_words)
    pad = 4 * ((batch + 3) // 4)
    g = gathered.view(world, n)
    obs = g[:, : batch * obs_words].reshape(world * batch, obs_words)
    reward = g[:, batch * obs_words : batch * (obs_words + 1)].contiguous().view(torch.float32).reshape(-1)
    flags = g[:, batch * (obs_words + 1) :].contiguous().view(torch.uint8).view(world, -1)
    return obs, reward, flags[:, :batch].reshape(-1), flags[:, pad : pad + batch].reshape(-1)


class OverlappedGather:
    """Double-buffered all-gather of a per-rank snapshot that overlaps with the work enqueued after it.

    `submit(fill)` lets `fill(buffer)` enqueue the snapshot (e.g. `env.observe_packed(out=buffer)`) on the
    current stream and hands the PREVIOUS snapshot to a side stream for the collective; `flush()` hands
    over the last one.  The hand-over goes through the host -- wait for the snapshot's event, then enqueue
    the collective -- not through a stream-to-stream event wait: on ROCm 7 / MI355X, once a second stream
    has waited on an event of the stepping stream, every later hipGraph replay on that stream runs ~40 %
    slower per kernel (tools/gather_probe.py), while the host-mediated form costs nothing measurable.
    The host therefore runs at most one snapshot ahead of the device.  CPU tensors (gloo) gather inline."""

    def __init__(self, shard_shape, dtype: torch.dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        shard_shape = tuple(shard_shape)
        self.snap = [torch.empty(shard_shape, dtype=dtype, device=self.device) for _ in range(2)]
        self.out = [torch.empty((self.world * shard_shape[0],) + shard_shape[1:], dtype=dtype, device=self.device) for _ in range(2)]
        self.n = 0
        self.pending: Optional[int] = None
        self.completed: List[int] = []  # buffers whose gather has been enqueued, oldest first
        if self.cuda:
            self.comm = torch.cuda.Stream(device=self.device)
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.done = [torch.cuda.Event() for _ in range(2)]

    def _hand_over(self, b: int):
        if self.cuda:
            self.ready[b].synchronize()
            with torch.cuda.stream(self.comm):
                all_gather_observation(self.snap[b], out=self.out[b], group=self.group)
                self.done[b].record(self.comm)
        else:
            all_gather_observation(self.snap[b], out=self.out[b], group=self.group)
        self.completed.append(b)
        del self.completed[:-2]

    def submit(self, fill: Callable[[torch.Tensor], None]):
        b = self.n & 1
        if self.cuda and self.n >= 2:
            self.done[b].synchronize()  # the gather that read snap[b] is over
        fill(self.snap[b])
        if self.cuda:
            self.ready[b].record(torch.cuda.current_stream(self.device))
        if self.pending is not None:
            self._hand_over(self.pending)
        self.pending = b
        self.n += 1

    def flush(self):
        if self.pending is not None:
            self._hand_over(self.pending)
            self.pending = None

    def latest(self) -> Optional[torch.Tensor]:
        """The most recently gathered `[W * B_local, ...]` tensor (waits for its collective), or None."""
        if not self.completed:
            return None
        b = self.completed[-1]
        if self.cuda:
            self.done[b].synchronize()
        return self.out[b]


class _DeviceBytes:
    """A device range owned by libqgym as a `__cuda_array_interface__` object (torch.as_tensor makes a view, no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def split_gathered(gathered: torch.Tensor, layout, world: int, word_bytes: int = 4):
    """`world` consecutive shards (uint8, `world * layout.bytes`) -> (obs [world*batch, words] of `word_bytes`-byte integers,
    reward [world*batch] f32, is_final, success [world*batch] uint8), rank order = env order.  `layout`: VecEnv.shard_layout()."""
    B = int(layout.batch)
    g = gathered.view(torch.uint8).view(world, int(layout.bytes))
    dt = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[word_bytes]
    obs = g[:, : int(layout.obs_bytes)].contiguous().view(dt).view(world * B, -1)
    ro, fo, so = int(layout.reward_offset), int(layout.final_offset), int(layout.success_offset)
    reward = g[:, ro : ro + 4 * B].contiguous().view(torch.float32).reshape(-1)
    return obs, reward, g[:, fo : fo + B].reshape(-1), g[:, so : so + B].reshape(-1)


class Communicator:
    """`qg_comm` of include/qgym.h: the multi-GPU hand-over below Python -- RCCL's all-gather called from libqgym, or the
    direct write into every peer's window over xGMI.  One per rank (process or thread), on the rank's GPU.

    `unique_id`: the 128 bytes rank 0 got from `Communicator.unique_id()`, moved to every rank by the host's own control channel
    (a file, a socket, a `torch.distributed` broadcast on gloo); `None` with `local=True` builds a communicator without RCCL, for
    the direct-write transport with handles exchanged by the host."""

    def __init__(self, rank: int, world: int, device: Optional[int] = None, unique_id: Optional[bytes] = None, local: bool = False):
        import ctypes as C

        from . import _lib

        self._lib, self._C = _lib, C
        self._L = _lib.load()
        self.rank, self.world = int(rank), int(world)
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        if local:
            _lib.check(self._L.qg_comm_init_local(self.rank, self.world, self.device_index, C.byref(h)))
        else:
            if unique_id is None or len(unique_id) != _lib.QG_COMM_ID_BYTES:
                raise ValueError("Communicator: pass the 128-byte id of Communicator.unique_id() (made on rank 0), or local=True")
            buf = (C.c_uint8 * _lib.QG_COMM_ID_BYTES).from_buffer_copy(unique_id)
            _lib.check(self._L.qg_comm_init(buf, self.rank, self.world, self.device_index, C.byref(h)))
        self._h = h
        self.shard_bytes = 0

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _lib

        buf = (C.c_uint8 * _lib.QG_COMM_ID_BYTES)()
        _lib.check(_lib.load().qg_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None):
            self._L.qg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _view(self, ptr, nbytes: int) -> torch.Tensor:
        return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=self.device)

    # ---- RCCL ---------------------------------------------------------------------------------------------------
    def gather(self, env, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Pack `env`'s shard and ncclAllGather it on the current stream -> uint8 [world * shard bytes]."""
        n = int(env.shard_layout().bytes) * self.world
        if out is None:
            out = torch.empty(n, dtype=torch.uint8, device=self.device)
        if out.device != self.device or out.numel() * out.element_size() != n or not out.is_contiguous():
            raise ValueError(f"gather: `out` must be a contiguous buffer of {n} bytes on the communicator's device")
        self._lib.check(self._L.qg_vec_gather_learner_shard(env._h, self._h, out.data_ptr(), self._stream()))
        return out

    def submit(self, env):
        """Snapshot `env`'s shard on the current stream; the previous snapshot goes to the all-gather on the side stream."""
        self.shard_bytes = int(env.shard_layout().bytes)
        self._lib.check(self._L.qg_comm_gather_submit(self._h, env._h, self._stream()))

    def flush(self):
        self._lib.check(self._L.qg_comm_gather_flush(self._h))

    def latest(self) -> Optional[torch.Tensor]:
        """The most recently gathered buffer (waits for its collective), or None.  A view into the communicator's memory."""
        p = self._C.c_void_p()
        self._lib.check(self._L.qg_comm_gather_latest(self._h, self._C.byref(p)))
        return self._view(p.value, self.shard_bytes * self.world) if p.value else None

    # ---- direct write -------------------------------------------------------------------------------------------
    def p2p_connect(self, shard_bytes: int):
        """Allocate the window, exchange the hipIpc handles over the communicator's RCCL, map the peers' windows (collective)."""
        self.shard_bytes = int(shard_bytes)
        self._lib.check(self._L.qg_comm_p2p_connect(self._h, self.shard_bytes))

    def p2p_export(self, shard_bytes: int) -> bytes:
        self.shard_bytes = int(shard_bytes)
        buf = (self._C.c_uint8 * self._lib.QG_P2P_HANDLE_BYTES)()
        self._lib.check(self._L.qg_comm_p2p_export(self._h, self.shard_bytes, buf))
        return bytes(buf)

    def p2p_open(self, handles):
        """handles: the `world` exported handles in rank order (this rank's own entry is not opened)."""
        blob = b"".join(handles)
        if len(blob) != self.world * self._lib.QG_P2P_HANDLE_BYTES:
            raise ValueError("p2p_open: one 64-byte handle per rank")
        buf = (self._C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self._lib.check(self._L.qg_comm_p2p_open(self._h, buf))

    def push(self, env):
        """Pack `env`'s shard and write it into every rank's window, then raise this rank's arrival flag there (current stream)."""
        self._lib.check(self._L.qg_vec_push_learner_shard(env._h, self._h, self._stream()))

    def wait(self) -> torch.Tensor:
        """Enqueue the wait for every rank's shard of the next epoch; returns the gathered view (valid for work enqueued after this
        call on the current stream, until `release`)."""
        p = self._C.c_void_p()
        self._lib.check(self._L.qg_comm_p2p_wait(self._h, self._C.byref(p), self._stream()))
        return self._view(p.value, self.shard_bytes * self.world)

    def release(self):
        self._lib.check(self._L.qg_comm_p2p_release(self._h, self._stream()))

    def check(self):
        """Synchronise the current stream and raise if a peer missed a deadline."""
        self._lib.check(self._L.qg_comm_p2p_check(self._h, self._stream()))

    def p2p_reset(self):
        """Restart the direct-write transport after an error (every rank, between two barriers of the host's own)."""
        self._lib.check(self._L.qg_comm_p2p_reset(self._h, self._stream()))


def unpack_rows_u32(packed: torch.Tensor, dim: int) -> torch.Tensor:
    """Bit-packed rows `[B, D]` (int32 words, bit c = entry (r, c)) -> dense int8 `[B, D, dim]`.
    Reference-layout helper for consumers of the gathered tensor (pure torch, any device)."""
    shifts = torch.arange(dim, device=packed.device, dtype=torch.int64)
    words = packed.to(torch.int64) & 0xFFFFFFFF
    return ((words.unsqueeze(-1) >> shifts) & 1).to(torch.int8)
