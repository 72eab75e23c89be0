"""VecEnv -- B independent synthesis environments resident on one MI355X.

The batched counterpart of the reference's scalar env objects (`qiskit_gym_rs.CliffordEnv` etc.,
reference rust/src/envs/clifford.rs:285-382): same method names, every method acting on the whole
batch, tensors in and out.  All state lives in HBM behind the C ABI (`include/qgym.h`); this class
only marshals pointers.  torch is used for device memory and streams, nothing else.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .envs.gateset import parse_gateset


def _stream_ptr(device=None) -> int:
    """Current torch stream of `device` (default: the current device) -- for the handle-free C entry points, which work on
    the calling thread's current HIP device.  Handle-bound calls use `VecEnv._stream()`."""
    return torch.cuda.current_stream(device).cuda_stream


class VecEnv:
    """`batch` copies of one env kind (`"clifford" | "linear_function" | "permutation" | "pauli"`)."""

    def __init__(self, env_kind: str, num_qubits: int, gateset: Sequence, batch: int, device: int | None = None,
                 metrics_weights: dict | None = None, seed: int | None = None, env_base: int = 0, **config):
        if not torch.cuda.is_available():
            raise RuntimeError("qiskit_gym_amd.VecEnv needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback")
        self._L = _lib.load()
        self.env_kind = env_kind
        self.num_qubits = int(num_qubits)
        self.gateset = [(g[0], tuple(int(q) for q in g[1])) for g in gateset]
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.batch = int(batch)
        self.config = dict(config)  # the constructor options as given (reference defaults where absent)
        self._cfg = _lib.make_config(env_kind, num_qubits, metrics_weights=metrics_weights, **config)
        self._gates = _lib.make_gates(parse_gateset(self.gateset))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.qg_vec_create(C.byref(self._cfg), self._gates, len(self.gateset), self.batch,
                                             self.device_index, C.byref(h)))
        self._h = h
        if seed is not None:  # the handle's own RNG streams (add_inverts coins, PauliEnv observe permutations)
            _lib.check(self._L.qg_vec_set_seed(self._h, int(seed) & (2**64 - 1)))
        if env_base:
            self.set_env_base(env_base)
        info = _lib.QGVecInfo()
        _lib.check(self._L.qg_vec_get_info(self._h, C.byref(info)))
        self.num_actions_ = info.num_actions
        self.obs_shape_ = (info.obs_rows, info.obs_cols)
        self.packed_word_bytes = info.packed_word_bytes
        self.packed_words_per_env = info.packed_words_per_env
        # result arrays live in torch tensors bound into the handle (zero-copy for the learner)
        self.reward = torch.empty(self.batch, dtype=torch.float32, device=self.device)
        self.done = torch.empty(self.batch, dtype=torch.uint8, device=self.device)
        self.success = torch.empty(self.batch, dtype=torch.uint8, device=self.device)
        self.depth = torch.empty(self.batch, dtype=torch.int32, device=self.device)
        _lib.check(self._L.qg_vec_bind_outputs(self._h, self.reward.data_ptr(), self.done.data_ptr(),
                                               self.success.data_ptr(), self.depth.data_ptr()))

    def _stream(self) -> int:
        """The current torch stream of THIS env's device (not of whatever device happens to be current)."""
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "_h", None):
            self._L.qg_vec_destroy(self._h)
            self._h = None
        self._dense = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Env trait, batched -------------------------------------------------------------------
    def num_actions(self) -> int:
        return self.num_actions_

    def obs_shape(self):
        return list(self.obs_shape_)

    @property
    def difficulty(self) -> int:
        return int(self._L.qg_vec_get_difficulty(self._h))

    @difficulty.setter
    def difficulty(self, d: int):
        _lib.check(self._L.qg_vec_set_difficulty(self._h, int(d)))

    def set_env_base(self, first_env: int):
        """This batch is the shard [first_env, first_env + batch) of a larger one: every counter-RNG draw of env e is made
        as env first_env + e, so a sharded (multi-GPU) run equals the unsharded run of the whole batch."""
        _lib.check(self._L.qg_vec_set_env_base(self._h, int(first_env)))

    def set_state(self, states, fmt: str = "i64"):
        """states: [B, n] tensor/array in the trait's `Vec<i64>` wire format (`fmt="i64"`), dense
        0/1 bytes (`"u8"`) or bit-packed rows (`"packed"`)."""
        code = {"i64": _lib.FMT_I64, "u8": _lib.FMT_U8, "packed": _lib.FMT_PACKED}[fmt]
        if isinstance(states, torch.Tensor) and states.is_cuda:
            if states.device != self.device:
                raise ValueError("set_state: the tensor must live on the env's device")
            size = {"i64": 8, "u8": 1}.get(fmt, self.packed_word_bytes)
            ok = {8: (torch.int64, torch.uint64), 4: (torch.int32, torch.uint32), 1: (torch.uint8, torch.int8)}[size]
            if states.dtype not in ok:
                raise TypeError(f"set_state(fmt={fmt!r}): a device tensor must have {size}-byte integer elements, got {states.dtype}")
            if states.numel() % self.batch:
                raise ValueError(f"set_state: {states.numel()} elements do not divide into {self.batch} envs")
            t = states.contiguous().view(self.batch, -1)
            _lib.check(self._L.qg_vec_set_state(self._h, t.data_ptr(), code, t.shape[1], 1, self._stream()))
            return
        dt = {"i64": np.int64, "u8": np.uint8}.get(fmt)
        if dt is None:
            dt = {1: np.uint8, 4: np.uint32, 8: np.uint64}[self.packed_word_bytes]
        a = np.ascontiguousarray(np.asarray(states.cpu() if isinstance(states, torch.Tensor) else states, dtype=dt)).reshape(self.batch, -1)
        _lib.check(self._L.qg_vec_set_state(self._h, a.ctypes.data, code, a.shape[1], 0, self._stream()))

    def get_state(self, fmt: str = "i64") -> torch.Tensor:
        code = {"i64": _lib.FMT_I64, "u8": _lib.FMT_U8, "packed": _lib.FMT_PACKED}[fmt]
        if self.env_kind == "permutation":
            n = self.num_qubits
        elif fmt == "packed":
            n = self.packed_words_per_env
        else:
            d = self.obs_shape_[0]
            n = d * d
        dt = {"i64": torch.int64, "u8": torch.uint8}.get(fmt)
        if dt is None:
            dt = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[self.packed_word_bytes]
        out = torch.empty((self.batch, n), dtype=dt, device=self.device)
        _lib.check(self._L.qg_vec_get_state(self._h, out.data_ptr(), code, n, 1, self._stream()))
        return out

    def reset(self, seed: int = 0):
        _lib.check(self._L.qg_vec_reset(self._h, int(seed) & (2**64 - 1), self._stream()))

    def reset_done(self, seed: int):
        """reset() only the envs whose episode is over (`done` set); stream-ordered, no host sync."""
        _lib.check(self._L.qg_vec_reset_done(self._h, int(seed) & (2**64 - 1), self._stream()))

    def reset_done_step(self, seed: int, actions: torch.Tensor, coins: Optional[torch.Tensor] = None,
                        rewards_out: Optional[torch.Tensor] = None, dones_out: Optional[torch.Tensor] = None):
        """`reset_done(seed)` then `step(actions)` -- same results -- as one launch where the layout allows it (`qg_vec_reset_done_step`)."""
        actions = actions.contiguous()
        if actions.numel() != self.batch:
            raise ValueError(f"reset_done_step: one action per env ({self.batch}), got {tuple(actions.shape)}")
        ptr, dt = self._act(actions)
        cp = None
        if coins is not None:
            coins = coins.to(device=self.device, dtype=torch.uint8).contiguous()
            cp = coins.data_ptr()
        for name, t, size in (("rewards_out", rewards_out, 4), ("dones_out", dones_out, 1)):
            if t is not None and (t.device != self.device or t.numel() != self.batch or t.element_size() != size or not t.is_contiguous()):
                raise ValueError(f"reset_done_step: {name} must be a contiguous [B] tensor of {size}-byte elements on the env's device")
        rp = rewards_out.data_ptr() if rewards_out is not None else None
        dp = dones_out.data_ptr() if dones_out is not None else None
        _lib.check(self._L.qg_vec_reset_done_step(self._h, int(seed) & (2**64 - 1), ptr, dt, cp, rp, dp, self._stream()))
        return self.reward, self.done

    def set_clock(self, clock: Optional[torch.Tensor]):
        """Attach (or detach with None) a device clock: an int64 [1] tensor on this device that every
        RNG-driven kernel of the handle adds to its counter (`qg_vec_set_clock`) -- what lets a captured
        hipGraph of resets / steps draw fresh randomness on every replay."""
        if clock is not None and (clock.dtype != torch.int64 or clock.numel() != 1 or clock.device != self.device):
            raise ValueError("clock must be an int64 tensor with one element on the env's device")
        _lib.check(self._L.qg_vec_set_clock(self._h, clock.data_ptr() if clock is not None else None))
        self._clock = clock  # keep it alive while attached

    def set_counters(self, step_index: int, observe_index: int = 0):
        """Host-side RNG counters of the next step (add_inverts coin) and the next PauliEnv observe()."""
        _lib.check(self._L.qg_vec_set_counters(self._h, int(step_index), int(observe_index)))

    def reset_with(self, actions: torch.Tensor):
        """actions: int32 [difficulty, B] scramble draws (the reference's reset() RNG made explicit)."""
        a = actions.to(device=self.device, dtype=torch.int32).contiguous().view(-1, self.batch)
        _lib.check(self._L.qg_vec_reset_with(self._h, a.data_ptr(), a.shape[0], self._stream()))

    def _act(self, actions: torch.Tensor) -> Tuple[int, int]:
        if actions.device != self.device:
            raise ValueError("actions must live on the env's device")
        if actions.dim() == 0 or actions.shape[-1] != self.batch:
            raise ValueError(f"actions must be [..., {self.batch}] (one per env), got {tuple(actions.shape)}")
        if actions.dtype == torch.int32:
            return actions.data_ptr(), _lib.ACT_I32
        if actions.dtype == torch.int64:
            return actions.data_ptr(), _lib.ACT_I64
        raise TypeError("actions must be int32 or int64")

    def step(self, actions: torch.Tensor, coins: Optional[torch.Tensor] = None):
        """One env.step() for every env: one kernel launch.  Returns (reward, done) views."""
        actions = actions.contiguous()
        if actions.numel() != self.batch:
            raise ValueError(f"step: one action per env ({self.batch}), got {tuple(actions.shape)}")
        ptr, dt = self._act(actions)
        cp = None
        if coins is not None:
            coins = coins.to(device=self.device, dtype=torch.uint8).contiguous()
            if coins.numel() != self.batch:
                raise ValueError("step: one coin per env")
            cp = coins.data_ptr()
        _lib.check(self._L.qg_vec_step(self._h, ptr, dt, cp, self._stream()))
        return self.reward, self.done

    def rollout(self, actions: torch.Tensor, fused: bool = False, coins: Optional[torch.Tensor] = None,
                rewards_out: Optional[torch.Tensor] = None, dones_out: Optional[torch.Tensor] = None):
        """actions [T, B].  fused=False: T single-step launches replayed from a hipGraph;
        fused=True: one launch with the state held in registers across the T steps."""
        actions = actions.contiguous()
        if actions.dim() != 2:
            raise ValueError(f"rollout: actions must be [T, {self.batch}], got {tuple(actions.shape)}")
        T = actions.shape[0]
        ptr, dt = self._act(actions)
        if coins is not None and coins.numel() != T * self.batch:
            raise ValueError("rollout: coins must be [T, B]")
        for name, t, size in (("rewards_out", rewards_out, 4), ("dones_out", dones_out, 1)):
            if t is not None and (t.device != self.device or t.numel() != T * self.batch or t.element_size() != size or not t.is_contiguous()):
                raise ValueError(f"rollout: {name} must be a contiguous [T, B] tensor of {size}-byte elements on the env's device")
        cp = None
        if coins is not None:
            coins = coins.to(device=self.device, dtype=torch.uint8).contiguous()
            cp = coins.data_ptr()
        rp = rewards_out.data_ptr() if rewards_out is not None else None
        dp = dones_out.data_ptr() if dones_out is not None else None
        _lib.check(self._L.qg_vec_rollout(self._h, ptr, dt, T, cp, rp, dp, 1 if fused else 0, self._stream()))
        return self.reward, self.done

    def rollout_ring(self, actions: torch.Tensor, n_steps: int):
        """n_steps single-step launches (one cached hipGraph); step t uses actions[t % len(actions)]."""
        actions = actions.contiguous()
        if actions.dim() != 2:
            raise ValueError(f"rollout_ring: actions must be [period, {self.batch}], got {tuple(actions.shape)}")
        ptr, dt = self._act(actions)
        _lib.check(self._L.qg_vec_rollout_ring(self._h, ptr, dt, int(n_steps), actions.shape[0], self._stream()))
        return self.reward, self.done

    def observe(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Dense int8 observation [B, rows, cols] (the Gym adapter's `_full_obs`, adapters.py:50-54)."""
        r, c = self.obs_shape_
        if out is None:
            out = torch.empty((self.batch, r, c), dtype=torch.int8, device=self.device)
        _lib.check(self._L.qg_vec_observe_dense(self._h, out.data_ptr(), self._stream()))
        return out

    def track_dense(self, enable: bool = True) -> Optional[torch.Tensor]:
        """Keep the dense int8 observation [B, rows, cols] resident and current (`qg_vec_track_dense`): the returned tensor equals
        `observe()` after every later call, in stream order; a step rewrites only the rows its gate changed.  `enable=False` detaches."""
        if not enable:
            _lib.check(self._L.qg_vec_track_dense(self._h, None, self._stream()))
            self._dense = None
            return None
        r, c = self.obs_shape_
        dense = torch.empty((self.batch, r, c), dtype=torch.int8, device=self.device)
        _lib.check(self._L.qg_vec_track_dense(self._h, dense.data_ptr(), self._stream()))
        self._dense = dense  # the handle writes into it until detached: keep it alive
        return dense

    _DTYPES = {torch.int8: _lib.QG_DT_I8, torch.float32: _lib.QG_DT_F32, torch.bfloat16: _lib.QG_DT_BF16, torch.float16: _lib.QG_DT_F16}

    def observe_as(self, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Dense observation [B, rows*cols] of {0, 1} written directly in `dtype` (what the policy's first layer reads)."""
        r, c = self.obs_shape_
        if out is None:
            out = torch.empty((self.batch, r * c), dtype=dtype, device=self.device)
        if out.dtype != dtype or out.numel() != self.batch * r * c or not out.is_contiguous():
            raise ValueError("observe_as: `out` must be a contiguous [B, rows*cols] tensor of the requested dtype")
        _lib.check(self._L.qg_vec_observe_dense_as(self._h, out.data_ptr(), self._DTYPES[dtype], self._stream()))
        return out

    def pauli_observe(self, perm_idx: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """PauliEnv observe() with the qubit-permutation draws given explicitly (int32 [B])."""
        r, c = self.obs_shape_
        if out is None:
            out = torch.empty((self.batch, r, c), dtype=torch.int8, device=self.device)
        pp = None
        if perm_idx is not None:
            perm_idx = perm_idx.to(device=self.device, dtype=torch.int32).contiguous()
            pp = perm_idx.data_ptr()
        _lib.check(self._L.qg_vec_pauli_observe_dense(self._h, out.data_ptr(), pp, self._stream()))
        return out

    def pauli_num_perms(self) -> int:
        return int(self._L.qg_vec_pauli_num_perms(self._h))

    def observe_packed(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Bit-packed observation [B, D] words (what the multi-GPU all-gather moves)."""
        dt = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[self.packed_word_bytes]
        if out is None:
            out = torch.empty((self.batch, self.packed_words_per_env), dtype=dt, device=self.device)
        _lib.check(self._L.qg_vec_observe_packed(self._h, out.data_ptr(), self._stream()))
        return out

    def shard_layout(self) -> "_lib.QGShardLayout":
        """Byte layout of this batch's learner shard: packed observation, f32 rewards, is_final and success bytes (qg_shard_layout)."""
        lay = _lib.QGShardLayout()
        _lib.check(self._L.qg_vec_learner_shard_layout(self._h, C.byref(lay)))
        return lay

    def pack_learner_shard(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """What the learner reads from every env after a step, as one flat uint8 buffer (`shard_layout()`)."""
        lay = self.shard_layout()
        if out is None:
            out = torch.empty(lay.bytes, dtype=torch.uint8, device=self.device)
        if out.device != self.device or out.numel() * out.element_size() != lay.bytes or not out.is_contiguous():
            raise ValueError(f"pack_learner_shard: `out` must be a contiguous buffer of {lay.bytes} bytes on the env's device")
        _lib.check(self._L.qg_vec_pack_learner_shard(self._h, out.data_ptr(), self._stream()))
        return out

    def masks(self) -> torch.Tensor:
        out = torch.empty((self.batch, self.num_actions_), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.qg_vec_masks(self._h, out.data_ptr(), self._stream()))
        return out

    def pauli_reset_from(self, tableaus: np.ndarray, labels: Sequence[Sequence[str]]):
        t = np.ascontiguousarray(np.asarray(tableaus, dtype=np.uint8).reshape(self.batch, -1))
        n_rot = np.ascontiguousarray(np.array([len(l) for l in labels], dtype=np.int32))
        blob = "".join("".join(l) for l in labels).encode()
        _lib.check(self._L.qg_vec_pauli_reset_from(self._h, t.ctypes.data, blob, n_rot.ctypes.data, self._stream()))

    def kernel_clock(self, n_slots: int) -> Optional[torch.Tensor]:
        """Diagnostics (`qg_vec_set_kernel_clock`): the next `n_slots` step / observation launches have every wave write its {entry, exit} device-clock
        ticks into the returned int64 tensor [n_slots, waves, 2] (zeroed); `n_slots=0` detaches.  See `kernel_durations_us`."""
        if n_slots <= 0:
            _lib.check(self._L.qg_vec_set_kernel_clock(self._h, None, 0, 0))
            self._kclk = None
            return None
        waves = max(8192, (self.batch + 7) // 8)  # the widest stamped grid: PauliEnv's reset trees, B / 32 workgroups of four waves
        slots = torch.zeros((n_slots, waves, 2), dtype=torch.int64, device=self.device)
        _lib.check(self._L.qg_vec_set_kernel_clock(self._h, slots.data_ptr(), n_slots, waves))
        self._kclk = slots
        return slots

    def kernel_durations_us(self, slots: torch.Tensor) -> np.ndarray:
        """Last wave exit - first wave entry of every stamped launch of `kernel_clock`'s tensor, in microseconds (slots no launch stamped are left out)."""
        rate = int(self._L.qg_kernel_clock_rate_khz(self.device_index))
        if rate <= 0:
            _lib.check(rate)
        t0, t1 = slots[..., 0], slots[..., 1]
        live = t1 != 0
        first = torch.where(live, t0, torch.full_like(t0, torch.iinfo(torch.int64).max)).amin(dim=1)
        last = t1.amax(dim=1)
        ok = live.any(dim=1)
        return ((last - first)[ok]).cpu().numpy().astype(np.float64) * 1e3 / rate

    def sync(self):
        """Wait for the current stream and raise if any env hit a fault the reference panics on."""
        _lib.check(self._L.qg_vec_sync(self._h, self._stream()))

    def solution(self, env: int):
        n = self._L.qg_vec_solution(self._h, env, None, 0)
        if n < 0:
            _lib.check(int(n))
        buf = (C.c_uint64 * max(int(n), 1))()
        self._L.qg_vec_solution(self._h, env, buf, int(n))
        return [int(buf[i]) for i in range(int(n))]

    def solutions(self, cap: Optional[int] = None):
        """Env::solution of every env at once (`qg_vec_solutions`): (entries uint64 [B, cap], lengths int64 [B]) as numpy arrays."""
        cap = int(self.config.get("max_depth", 128) if cap is None else cap)
        out = np.zeros((self.batch, max(cap, 1)), dtype=np.uint64)
        lens = np.zeros(self.batch, dtype=np.int64)
        _lib.check(self._L.qg_vec_solutions(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap, lens.ctypes.data_as(C.POINTER(C.c_int64))))
        return out[:, :cap], lens
