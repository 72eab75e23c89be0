"""GPU-resident rollout collection: observation, policy forward, sampling, env.step, auto-reset and
advantage estimation all stay on the device.

The reference collects episodes inside twisterl: rayon workers clone the scalar env per episode and
run a Rust copy of the policy on the CPU (`rl/synthesis.py:128-138`, notebook timing keys `collect`,
`data_to_torch`).  With the env batch resident in HBM one collection step is

    reset_done -> observe (bits -> policy dtype) -> 3 GEMMs -> sample -> step

i.e. five libqgym launches plus the policy's GEMMs, and the trajectories are born as torch tensors --
there is no `data_to_torch`.  The observation is written straight in the dtype the first layer
reads (`qg_vec_observe_dense_as`), actions / log-probs / entropy / values come out of one sampling
kernel (`qg_sample_actions`), rewards and episode ends are written by the step kernel into row t of
the rollout, and GAE(lambda) runs as one kernel over the finished [T, B] rollout (`qg_gae`).

For the reference's default policy shape in bf16 (and batches of a few thousand envs or more) the forward
pass needs no tensor library and no observation at all:

    reset_done -> [packed observation for the rollout] -> first layer from the bits -> middle layer + head + draw -> step

`qg_vec_embed` reads the resident TILE-layout state, `qg_policy_embed_words` the packed observation words of every
other layout (PauliEnv, wide CliffordEnv), `qg_policy_mid_head_sample` does the rest in one kernel.

`BasicPolicy` mirrors the shape of the reference's default policy network (`twisterl.nn.BasicPolicy`
as configured by `BasicPolicyConfig`, `rl/configs.py:531-607`; checkpoint shapes in
`examples/models/*.pt`): Linear(prod(obs_shape) -> 512) -> ReLU -> Linear(512 -> 256) -> ReLU ->
{Linear(256 -> num_actions), Linear(256 -> 1)}.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .vec import VecEnv, _stream_ptr

_DT = {torch.float32: _lib.QG_DT_F32, torch.bfloat16: _lib.QG_DT_BF16, torch.float16: _lib.QG_DT_F16, torch.int8: _lib.QG_DT_I8}


def sample_actions(logits: torch.Tensor, seed: int, counter: int, num_actions: Optional[int] = None, mask: Optional[torch.Tensor] = None,
                   value_col: Optional[int] = None, actions: Optional[torch.Tensor] = None, logp: Optional[torch.Tensor] = None,
                   entropy: Optional[torch.Tensor] = None, values: Optional[torch.Tensor] = None, clock: Optional[torch.Tensor] = None):
    """One categorical draw per row of `logits[:, :num_actions]` (f32 / bf16 / f16, row stride free).
    `clock`: optional int64 device scalar added to `counter` on the device (graph replays).
    Returns (actions int64, logp f32, entropy f32, values f32 or None); see `qg_sample_actions`."""
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("logits must be [B, >=num_actions] with unit column stride")
    B, dev = logits.shape[0], logits.device
    A = int(num_actions) if num_actions is not None else logits.shape[1]
    actions = torch.empty(B, dtype=torch.int64, device=dev) if actions is None else actions
    logp = torch.empty(B, dtype=torch.float32, device=dev) if logp is None else logp
    entropy = torch.empty(B, dtype=torch.float32, device=dev) if entropy is None else entropy
    if value_col is not None and values is None:
        values = torch.empty(B, dtype=torch.float32, device=dev)
    if mask is not None:
        mask = mask.to(torch.uint8).contiguous()
    act_dt = {torch.int32: _lib.ACT_I32, torch.int64: _lib.ACT_I64}[actions.dtype]
    L = _lib.load()
    _lib.check(L.qg_sample_actions(
        logits.data_ptr(), _DT[logits.dtype], logits.stride(0), B, A, mask.data_ptr() if mask is not None else None,
        int(seed) & (2**64 - 1), int(counter), clock.data_ptr() if clock is not None else None, actions.data_ptr(), act_dt, logp.data_ptr(), entropy.data_ptr(),
        -1 if value_col is None else int(value_col), values.data_ptr() if value_col is not None else None, _stream_ptr()))
    return actions, logp, entropy, values


def gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor, last_values: Optional[torch.Tensor], gamma: float,
        gae_lambda: float, advantages: Optional[torch.Tensor] = None, returns: Optional[torch.Tensor] = None):
    """GAE(lambda) over a [T, B] rollout; `dones[t]` = the episode ended with step t.  Returns (advantages, returns)."""
    T, B = rewards.shape
    for x, dt in ((rewards, torch.float32), (values, torch.float32), (dones, torch.uint8)):
        if x.shape != (T, B) or x.dtype != dt or not x.is_contiguous():
            raise ValueError("gae: rewards/values must be contiguous f32 [T, B], dones uint8 [T, B]")
    advantages = torch.empty_like(rewards) if advantages is None else advantages
    returns = torch.empty_like(rewards) if returns is None else returns
    lv = None
    if last_values is not None:
        lv = last_values.to(torch.float32).contiguous()
    _lib.check(_lib.load().qg_gae(rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), lv.data_ptr() if lv is not None else None,
                                  float(gamma), float(gae_lambda), T, B, advantages.data_ptr(), returns.data_ptr(), _stream_ptr()))
    return advantages, returns


def expand_packed(packed: torch.Tensor, cols: int, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Bit-packed observation rows (`VecEnv.observe_packed`, any leading shape) -> dense {0,1} [..., rows, cols] in `dtype`."""
    packed = packed.contiguous()
    if out is None:
        out = torch.empty((*packed.shape, cols), dtype=dtype, device=packed.device)
    _lib.check(_lib.load().qg_expand_packed(packed.data_ptr(), packed.element_size(), packed.numel(), int(cols), out.data_ptr(), _DT[dtype],
                                            _stream_ptr()))
    return out


def pack_embedding(env: VecEnv, weight: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """First-layer weight [hidden, rows*cols] (f32 / bf16) in the k order `embed` consumes (`qg_vec_pack_embedding`);
    repack after every optimiser step (pass `out` to rewrite the same buffer: graph-safe)."""
    if weight.dim() != 2 or weight.stride(1) != 1:
        raise ValueError("weight must be [hidden, obs_size] with unit column stride")
    L = _lib.load()
    hidden = weight.shape[0]
    nbytes = L.qg_vec_embed_packed_bytes(env._h, hidden)
    if nbytes == 0:
        raise ValueError("embed needs a TILE-layout env (CliffordEnv N <= 16, LinearFunctionEnv 8 < N <= 32) and hidden % 64 == 0")
    if out is None:
        out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
    _lib.check(L.qg_vec_pack_embedding(env._h, weight.data_ptr(), _DT[weight.dtype], weight.stride(0), hidden, out.data_ptr(), env._stream()))
    return out


def embed(env: VecEnv, packed: torch.Tensor, bias: Optional[torch.Tensor], hidden: int, relu: bool = True, out: Optional[torch.Tensor] = None,
          obs_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(obs @ W.T + bias) in bf16 [B, hidden], computed from the env's resident bit-packed state (`qg_vec_embed`).  `obs_out`: also
    `env.observe_packed(out=obs_out)` of the same state (`qg_vec_embed_observe`: one launch for small batches)."""
    if out is None:
        out = torch.empty((env.batch, hidden), dtype=torch.bfloat16, device=env.device)
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous f32 vector")
    L = _lib.load()
    if obs_out is not None:
        if (not obs_out.is_contiguous() or obs_out.numel() != env.batch * env.packed_words_per_env or obs_out.element_size() != env.packed_word_bytes
                or obs_out.device != out.device):
            raise ValueError("obs_out must be a contiguous [batch, packed_words_per_env] tensor of the env's packed word type on its device")
        _lib.check(L.qg_vec_embed_observe(env._h, packed.data_ptr(), bias.data_ptr() if bias is not None else None, hidden, int(relu), out.data_ptr(),
                                          out.stride(0), obs_out.data_ptr(), env._stream()))
        return out
    _lib.check(L.qg_vec_embed(env._h, packed.data_ptr(), bias.data_ptr() if bias is not None else None, hidden, int(relu), out.data_ptr(),
                              out.stride(0), env._stream()))
    return out


def pack_embed_words(weight: torch.Tensor, rows: int, cols: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """First-layer weight [hidden, rows*cols] (f32 / bf16) in the order `embed_words` consumes (`qg_policy_pack_embed_words`)."""
    if weight.dim() != 2 or weight.stride(1) != 1:
        raise ValueError("weight must be [hidden, rows*cols] with unit column stride")
    L = _lib.load()
    nbytes = L.qg_policy_embed_words_packed_bytes(int(rows), int(cols), weight.shape[0])
    if nbytes == 0:
        raise ValueError("embed_words needs an even number of rows, cols <= 64 and hidden % 128 == 0")
    if out is None:
        out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
    _lib.check(L.qg_policy_pack_embed_words(weight.data_ptr(), _DT[weight.dtype], weight.stride(0), int(rows), int(cols), weight.shape[0], out.data_ptr(),
                                            _stream_ptr()))
    return out


def embed_words(words: torch.Tensor, cols: int, packed: torch.Tensor, bias: Optional[torch.Tensor], hidden: int, relu: bool = True,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(obs @ W.T + bias) in bf16 [B, hidden] from packed observation rows [B, rows] int64 (`VecEnv.observe_packed` of a handle with
    64-bit row words, a packed rollout buffer, a gathered shard): `qg_policy_embed_words`."""
    if words.dim() != 2 or words.dtype != torch.int64 or not words.is_contiguous():
        raise ValueError("words must be a contiguous int64 [B, rows] tensor")
    B, rows = words.shape
    if out is None:
        out = torch.empty((B, hidden), dtype=torch.bfloat16, device=words.device)
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous f32 vector")
    _lib.check(_lib.load().qg_policy_embed_words(words.data_ptr(), B, rows, int(cols), packed.data_ptr(), bias.data_ptr() if bias is not None else None, hidden,
                                                 int(relu), out.data_ptr(), out.stride(0), _stream_ptr()))
    return out


def pack_head(weight: torch.Tensor, bias: Optional[torch.Tensor], num_actions: int, value_row: int, out: Optional[torch.Tensor] = None,
              after_mid: bool = False) -> torch.Tensor:
    """Last-layer weight [rows, in_features] (+ bias [rows]; f32 or bf16, same dtype) in the order `head_sample` (or, with
    after_mid, `mid_head_sample`) consumes: rows 0..num_actions-1 are the actions, row `value_row` the value head (`qg_policy_pack_head`)."""
    L = _lib.load()
    nbytes = L.qg_policy_head_packed_bytes(num_actions, weight.shape[1])
    if nbytes == 0:
        raise ValueError("fused head needs num_actions <= 222 and in_features % 64 == 0, <= 512")
    if weight.stride(1) != 1 or (bias is not None and (bias.dtype != weight.dtype or not bias.is_contiguous())):
        raise ValueError("weight must have unit column stride; bias contiguous and of the same dtype")
    if out is None:
        out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
    _lib.check(L.qg_policy_pack_head(weight.data_ptr(), bias.data_ptr() if bias is not None else None, _DT[weight.dtype], weight.stride(0), weight.shape[1],
                                     int(num_actions), int(value_row), int(after_mid), out.data_ptr(), _stream_ptr()))
    return out


def pack_mid(weight: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Middle-layer weight [256, in_features] (+ bias) in the order `mid_head_sample` consumes (`qg_policy_pack_mid`)."""
    L = _lib.load()
    nbytes = L.qg_policy_mid_packed_bytes(weight.shape[1], weight.shape[0])
    if nbytes == 0:
        raise ValueError("fused middle layer needs 256 output features and in_features % 32 == 0, <= 2048")
    if weight.stride(1) != 1 or (bias is not None and (bias.dtype != weight.dtype or not bias.is_contiguous())):
        raise ValueError("weight must have unit column stride; bias contiguous and of the same dtype")
    if out is None:
        out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
    _lib.check(L.qg_policy_pack_mid(weight.data_ptr(), bias.data_ptr() if bias is not None else None, _DT[weight.dtype], weight.stride(0), weight.shape[1],
                                    weight.shape[0], out.data_ptr(), _stream_ptr()))
    return out


def mid_head_sample(h: torch.Tensor, packed_mid: torch.Tensor, mid_features: int, packed_head: torch.Tensor, num_actions: int, seed: int, counter: int,
                    actions: Optional[torch.Tensor] = None, logp: Optional[torch.Tensor] = None, entropy: Optional[torch.Tensor] = None,
                    values: Optional[torch.Tensor] = None, clock: Optional[torch.Tensor] = None):
    """relu(h W2^T + b2) -> last layer -> categorical draw in one kernel (`qg_policy_mid_head_sample`); the head packed with after_mid=True."""
    if h.dim() != 2 or h.dtype != torch.bfloat16 or h.stride(1) != 1:
        raise ValueError("h must be bf16 [B, in_features] with unit column stride")
    B, dev = h.shape[0], h.device
    actions = torch.empty(B, dtype=torch.int64, device=dev) if actions is None else actions
    logp = torch.empty(B, dtype=torch.float32, device=dev) if logp is None else logp
    entropy = torch.empty(B, dtype=torch.float32, device=dev) if entropy is None else entropy
    values = torch.empty(B, dtype=torch.float32, device=dev) if values is None else values
    act_dt = {torch.int32: _lib.ACT_I32, torch.int64: _lib.ACT_I64}[actions.dtype]
    _lib.check(_lib.load().qg_policy_mid_head_sample(h.data_ptr(), h.stride(0), B, h.shape[1], packed_mid.data_ptr(), int(mid_features), packed_head.data_ptr(),
                                                     int(num_actions), int(seed) & (2**64 - 1), int(counter), clock.data_ptr() if clock is not None else None,
                                                     actions.data_ptr(), act_dt, logp.data_ptr(), entropy.data_ptr(), values.data_ptr(), _stream_ptr()))
    return actions, logp, entropy, values


def mid_head_sample_step(env: VecEnv, h: torch.Tensor, packed_mid: torch.Tensor, mid_features: int, packed_head: torch.Tensor, seed: int, counter: int,
                         actions: torch.Tensor, logp: torch.Tensor, entropy: torch.Tensor, values: torch.Tensor,
                         rewards: Optional[torch.Tensor] = None, dones: Optional[torch.Tensor] = None, reset_seed: Optional[int] = None):
    """`mid_head_sample` followed, in the same launch, by `env.step(actions)` and by the compaction of the finished envs for the next
    `env.reset_done` (`qg_vec_mid_head_sample_step`): same draws, same env results, two launches less per collection step.  The env's
    device clock (`env.set_clock`) is the sampling clock.  `reset_seed`: also `env.reset_done(reset_seed)` (`..._step_reset`: inside the
    same launch for small batches)."""
    if h.dim() != 2 or h.dtype != torch.bfloat16 or h.stride(1) != 1 or h.shape[0] != env.batch:
        raise ValueError("h must be bf16 [env.batch, in_features] with unit column stride")
    act_dt = {torch.int32: _lib.ACT_I32, torch.int64: _lib.ACT_I64}[actions.dtype]
    args = (env._h, h.data_ptr(), h.stride(0), h.shape[1], packed_mid.data_ptr(), int(mid_features), packed_head.data_ptr(), int(seed) & (2**64 - 1),
            int(counter), actions.data_ptr(), act_dt, logp.data_ptr(), entropy.data_ptr(), values.data_ptr(),
            rewards.data_ptr() if rewards is not None else None, dones.data_ptr() if dones is not None else None)
    if reset_seed is None:
        _lib.check(_lib.load().qg_vec_mid_head_sample_step(*args, env._stream()))
    else:
        _lib.check(_lib.load().qg_vec_mid_head_sample_step_reset(*args, int(reset_seed) & (2**64 - 1), env._stream()))
    return actions, logp, entropy, values


def head_sample(h: torch.Tensor, packed: torch.Tensor, num_actions: int, seed: int, counter: int, actions: Optional[torch.Tensor] = None,
                logp: Optional[torch.Tensor] = None, entropy: Optional[torch.Tensor] = None, values: Optional[torch.Tensor] = None,
                clock: Optional[torch.Tensor] = None):
    """Last layer + categorical draw in one kernel (`qg_policy_head_sample`): h bf16 [B, in_features] -> (actions, logp, entropy, values)."""
    if h.dim() != 2 or h.dtype != torch.bfloat16 or h.stride(1) != 1:
        raise ValueError("h must be bf16 [B, in_features] with unit column stride")
    B, dev = h.shape[0], h.device
    actions = torch.empty(B, dtype=torch.int64, device=dev) if actions is None else actions
    logp = torch.empty(B, dtype=torch.float32, device=dev) if logp is None else logp
    entropy = torch.empty(B, dtype=torch.float32, device=dev) if entropy is None else entropy
    values = torch.empty(B, dtype=torch.float32, device=dev) if values is None else values
    act_dt = {torch.int32: _lib.ACT_I32, torch.int64: _lib.ACT_I64}[actions.dtype]
    _lib.check(_lib.load().qg_policy_head_sample(h.data_ptr(), h.stride(0), B, h.shape[1], packed.data_ptr(), int(num_actions), int(seed) & (2**64 - 1),
                                                 int(counter), clock.data_ptr() if clock is not None else None, actions.data_ptr(), act_dt,
                                                 logp.data_ptr(), entropy.data_ptr(), values.data_ptr(), _stream_ptr()))
    return actions, logp, entropy, values


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

    @torch.no_grad()
    def fused_heads(self) -> Tuple[torch.Tensor, torch.Tensor, int]:
        """Policy and value heads as ONE weight [pad8(A + 1), common] / bias, so a collection step runs
        three GEMMs; column A of the output is the value.  Rebuild after every optimiser step."""
        A = self.policy_head.out_features
        rows = (A + 1 + 7) // 8 * 8
        w = torch.zeros((rows, self.common.out_features), dtype=self.policy_head.weight.dtype, device=self.policy_head.weight.device)
        b = torch.zeros(rows, dtype=w.dtype, device=w.device)
        w[:A], w[A] = self.policy_head.weight, self.value_head.weight[0]
        b[:A], b[A] = self.policy_head.bias, self.value_head.bias[0]
        return w, b, A


def _linear_relu(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    try:  # bias + ReLU in the GEMM epilogue (hipBLASLt)
        return torch._addmm_activation(b, x, w.t())
    except Exception:  # pragma: no cover - older torch
        return torch.relu_(torch.addmm(b, x, w.t()))


@dataclass
class Rollout:
    obs: torch.Tensor         # [T, B, rows*cols] int8 dense observation before each action, or
                              # [T, B, words] bit-packed rows when the collector stores packed observations
    actions: torch.Tensor     # [T, B] int64
    logp: torch.Tensor        # [T, B] float32 log-prob of the action under the collecting policy
    entropy: torch.Tensor     # [T, B] float32
    values: torch.Tensor      # [T, B] float32
    rewards: torch.Tensor     # [T, B] float32
    dones: torch.Tensor       # [T, B] uint8 (episode ended with this step)
    advantages: torch.Tensor  # [T, B] float32 GAE(lambda)
    returns: torch.Tensor     # [T, B] float32 advantages + values
    last_values: Optional[torch.Tensor] = None  # [B] float32 value of the state after the last step (GAE bootstrap)
    obs_packed: bool = False
    obs_cols: int = 0

    def dense_obs(self, dtype: torch.dtype = torch.float32, t: Optional[slice] = None) -> torch.Tensor:
        """[T', B, rows*cols] dense observation in `dtype` (re-expanded from the packed rows when stored packed)."""
        o = self.obs if t is None else self.obs[t]
        if not self.obs_packed:
            return o.to(dtype)
        return expand_packed(o, self.obs_cols, dtype).flatten(-2)


class RolloutCollector:
    """Steps `env` for T steps under `policy`, resetting finished episodes on the device.

    store_obs: "dense" keeps the int8 observation of every step (the Gym adapter's format);
    "packed" keeps the bit-packed rows (8x smaller for CliffordGym, one 64-bit word per observation row for
    PauliGym; `Rollout.dense_obs` expands them on demand).

    use_graph: capture the whole T-step collection (every env kernel, the policy GEMMs, sampling,
    GAE) into one hipGraph on the first call and replay it afterwards -- one host call per rollout
    instead of ~10 per step.  The rollout tensors are then reused between calls (clone what must
    outlive the next `collect`), and policy parameters must be updated in place (optimisers do).

    Randomness: step number n of the collector (counted over its lifetime, kept in a device clock
    so that graph replays advance it) resets finished episodes with seed `seed + 0x9E3779B9 * (n + 1)`
    and samples with counter n; both modes therefore produce the same trajectories."""

    def __init__(self, env: VecEnv, policy: nn.Module, dtype: torch.dtype = torch.bfloat16, seed: int = 0, gamma: float = 0.995,
                 gae_lambda: float = 0.995, store_obs: str = "dense", use_graph: bool = False, use_bit_embedding: Optional[bool] = None,
                 use_fused_head: Optional[bool] = None, use_fused_step: Optional[bool] = None):
        self.env = env
        self.policy = policy.to(device=env.device, dtype=dtype)
        self.dtype = dtype
        self.seed = int(seed)
        self.gamma, self.gae_lambda = float(gamma), float(gae_lambda)  # rl/configs.py:136-137
        r, c = env.obs_shape_
        self.obs_size = r * c
        if store_obs not in ("dense", "packed"):
            raise ValueError("store_obs must be 'dense' or 'packed'")
        if store_obs == "packed" and env.env_kind == "pauli" and env.obs_shape_[1] > 64:
            raise ValueError("PauliGym observations wider than 64 columns have no packed form")
        self.store_obs = store_obs
        self.use_graph = bool(use_graph)
        self._x = torch.empty((env.batch, self.obs_size), dtype=dtype, device=env.device)  # policy input
        self.clock = torch.zeros(1, dtype=torch.int64, device=env.device)  # collector steps taken so far
        env.set_clock(self.clock)
        # the policy-layer kernels at every batch size: below ~ 4 096 envs the launches take their small-batch shapes (a 32-env tile per
        # wave / workgroup, weight fragments straight from L2: CliffordGym 16q x 1 024 envs 19 us per step against 42 us on library GEMMs,
        # PauliGym 20q 36 against 44)
        if use_bit_embedding is None:
            use_bit_embedding = True
        if use_fused_head is None:
            use_fused_head = True
        self._heads = None
        self._embed = None  # (packed first-layer weight, f32 bias): the first layer reads the env's bits directly
        if isinstance(self.policy, BasicPolicy):
            w, b, A = self.policy.fused_heads()
            self._heads = (w, b, A)
            emb = self.policy.embeddings
            if use_bit_embedding and dtype == torch.bfloat16 and emb.out_features % 64 == 0 and emb.in_features == self.obs_size:
                try:
                    self._embed = (pack_embedding(env, emb.weight), emb.bias.detach().float().contiguous())
                except (ValueError, _lib.QGymError):
                    self._embed = None  # layouts without the bit-consuming kernel keep the dense first layer
        # layouts the TILE kernel does not cover (PauliEnv, CliffordEnv N > 16, ...): the same first layer from the packed observation
        # words the rollout stores anyway (64-bit row words, even row count)
        self._embed_words = None
        self._cur_words = None
        if (self._embed is None and use_bit_embedding and isinstance(self.policy, BasicPolicy) and dtype == torch.bfloat16 and store_obs == "packed"
                and env.packed_word_bytes == 8 and env.packed_words_per_env == r and self.policy.embeddings.in_features == self.obs_size):
            try:
                self._embed_words = (pack_embed_words(self.policy.embeddings.weight, r, c), self.policy.embeddings.bias.detach().float().contiguous())
            except ValueError:
                self._embed_words = None
        self._h1 = None
        self._obs_row = None  # the rollout row the next first-layer launch also fills with the packed observation
        self._head = None  # packed last layer for the fused head + sampling kernel (bf16 BasicPolicy within its limits)
        self._mid = None   # packed middle layer: then middle layer + head + sampling are ONE kernel and only the first layer stays outside
        if self._heads is not None and dtype == torch.bfloat16 and use_fused_head:
            w, b, A = self._heads
            try:
                try:
                    self._mid = pack_mid(self.policy.common.weight, self.policy.common.bias)
                except ValueError:
                    self._mid = None
                self._head = pack_head(w, b, A, A, after_mid=self._mid is not None)
                self._scratch_actions = torch.empty(env.batch, dtype=torch.int64, device=env.device)
                self._scratch_f32 = torch.empty((3, env.batch), dtype=torch.float32, device=env.device)
            except ValueError:
                self._head = self._mid = None
        # sampling, env.step and the reset of the finished envs in ONE call (qg_vec_mid_head_sample_step_reset): TILE-layout envs
        inverts = bool(env.config.get("add_inverts", True))
        can_fuse_step = (self._mid is not None and ((env.env_kind == "clifford" and env.num_qubits <= 16)  # with add_inverts too (qm_inv2_body)
                                                    or (env.env_kind == "linear_function" and 8 < env.num_qubits <= 32 and not inverts)))
        self._fused_step = can_fuse_step if use_fused_step is None else bool(use_fused_step) and can_fuse_step
        self._graph = None
        self._graph_T = 0
        self._graph_ro: Optional[Rollout] = None

    @property
    def steps_done(self) -> int:
        return int(self.clock.item())

    def _alloc(self, T: int) -> Rollout:
        env, B, dev = self.env, self.env.batch, self.env.device
        f32 = dict(dtype=torch.float32, device=dev)
        if self.store_obs == "packed":
            dt = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[env.packed_word_bytes]
            obs = torch.empty((T, B, env.packed_words_per_env), dtype=dt, device=dev)
        else:
            obs = torch.empty((T, B, self.obs_size), dtype=torch.int8, device=dev)
        return Rollout(obs=obs, actions=torch.empty((T, B), dtype=torch.int64, device=dev), logp=torch.empty((T, B), **f32),
                       entropy=torch.empty((T, B), **f32), values=torch.empty((T, B), **f32), rewards=torch.empty((T, B), **f32),
                       dones=torch.empty((T, B), dtype=torch.uint8, device=dev), advantages=torch.empty((T, B), **f32),
                       returns=torch.empty((T, B), **f32), obs_packed=self.store_obs == "packed", obs_cols=env.obs_shape_[1])

    def _observe(self, ro: Rollout, t: int):
        """Observation of the current state into row t of the rollout and, as `dtype`, into the policy input
        (not needed when the first layer consumes the bits directly)."""
        env = self.env
        if ro.obs_packed:
            if self._embed is not None:
                self._obs_row = ro.obs[t]  # written by the first layer's launch (qg_vec_embed_observe)
                return
            env.observe_packed(out=ro.obs[t])
            if self._embed_words is not None:
                self._cur_words = ro.obs[t]
            elif env.env_kind == "pauli":
                # PauliEnv.observe() with add_perms draws a permutation (pauli.rs:657-662): observe once, expand the stored row words
                expand_packed(ro.obs[t], env.obs_shape_[1], self.dtype, out=self._x.view(env.batch, *env.obs_shape_))
            else:
                env.observe_as(self.dtype, out=self._x)
        else:
            env.observe(out=ro.obs[t].view(env.batch, *env.obs_shape_))
            if self._embed is not None:
                pass
            elif env.env_kind == "pauli":
                # PauliEnv.observe() with add_perms draws a permutation (pauli.rs:657-662): observe once
                _lib.check(_lib.load().qg_widen_dense(ro.obs[t].data_ptr(), ro.obs[t].numel(), self._x.data_ptr(), _DT[self.dtype], _stream_ptr()))
            else:
                env.observe_as(self.dtype, out=self._x)

    def _refresh_heads(self):
        """Copy the policy / value head parameters into the fused head (in place: graph-safe)."""
        if self._heads is None:
            return
        w, b, A = self._heads
        pol = self.policy
        w[:A].copy_(pol.policy_head.weight)
        w[A].copy_(pol.value_head.weight[0])
        b[:A].copy_(pol.policy_head.bias)
        b[A : A + 1].copy_(pol.value_head.bias)
        if self._embed is not None:
            pack_embedding(self.env, pol.embeddings.weight, out=self._embed[0])
            self._embed[1].copy_(pol.embeddings.bias)
        if self._embed_words is not None:
            pack_embed_words(pol.embeddings.weight, *self.env.obs_shape_, out=self._embed_words[0])
            self._embed_words[1].copy_(pol.embeddings.bias)
        if self._head is not None:
            pack_head(w, b, A, A, out=self._head, after_mid=self._mid is not None)
        if self._mid is not None:
            pack_mid(pol.common.weight, pol.common.bias, out=self._mid)

    def _first_layer(self) -> torch.Tensor:
        pol = self.policy
        if self._embed is not None:
            if self._h1 is None:
                self._h1 = torch.empty((self.env.batch, pol.embeddings.out_features), dtype=self.dtype, device=self.env.device)
            obs_row, self._obs_row = self._obs_row, None
            return embed(self.env, self._embed[0], self._embed[1], pol.embeddings.out_features, relu=True, out=self._h1, obs_out=obs_row)
        if self._embed_words is not None:
            if self._h1 is None:
                self._h1 = torch.empty((self.env.batch, pol.embeddings.out_features), dtype=self.dtype, device=self.env.device)
            return embed_words(self._cur_words, self.env.obs_shape_[1], self._embed_words[0], self._embed_words[1], pol.embeddings.out_features, relu=True,
                               out=self._h1)
        return _linear_relu(self._x, pol.embeddings.weight, pol.embeddings.bias)

    def _fused_tail(self, counter: int, actions, logp, entropy, values, clock):
        """Everything after the first layer in one kernel (or, without the packed middle layer, GEMM + head kernel)."""
        A = self._heads[2]
        if self._mid is not None:
            mid_head_sample(self._first_layer(), self._mid, self.policy.common.out_features, self._head, A, self.seed, counter, actions=actions, logp=logp,
                            entropy=entropy, values=values, clock=clock)
        else:
            head_sample(self._body_layers(), self._head, A, self.seed, counter, actions=actions, logp=logp, entropy=entropy, values=values, clock=clock)

    def _body_layers(self) -> torch.Tensor:
        """The two hidden layers: [B, common] activations."""
        pol = self.policy
        return _linear_relu(self._first_layer(), pol.common.weight, pol.common.bias)

    def _trunk(self) -> torch.Tensor:
        """Fused-head forward of the current observation: [B, pad8(A + 1)], column A is the value."""
        w, b, _ = self._heads
        return torch.addmm(b, self._body_layers(), w.t())

    def _forward_sample(self, ro: Rollout, t: int):
        if self._head is not None:  # (middle layer +) last layer + draw in one kernel, h2 / the logits never reach memory
            self._fused_tail(t, ro.actions[t], ro.logp[t], ro.entropy[t], ro.values[t], self.clock)
            return
        if self._heads is not None:
            A = self._heads[2]
            sample_actions(self._trunk(), self.seed, t, num_actions=A, value_col=A, actions=ro.actions[t], logp=ro.logp[t],
                           entropy=ro.entropy[t], values=ro.values[t], clock=self.clock)
            return
        logits, value = self.policy(self._x)
        sample_actions(logits.contiguous(), self.seed, t, actions=ro.actions[t], logp=ro.logp[t], entropy=ro.entropy[t], clock=self.clock)
        ro.values[t].copy_(value)

    def _value_of_current_state(self) -> torch.Tensor:
        if self._embed_words is not None:
            if getattr(self, "_words_scratch", None) is None:
                self._words_scratch = torch.empty((self.env.batch, self.env.packed_words_per_env), dtype=torch.int64, device=self.env.device)
            self._cur_words = self.env.observe_packed(out=self._words_scratch)
        elif self._embed is None:
            self.env.observe_as(self.dtype, out=self._x)
        if self._head is not None:  # the fused kernel's value output (its draw is discarded)
            self._fused_tail(0, self._scratch_actions, self._scratch_f32[0], self._scratch_f32[1], self._scratch_f32[2], None)
            return self._scratch_f32[2]
        if self._heads is not None:
            return self._trunk()[:, self._heads[2]].float()
        return self.policy(self._x)[1].float()

    def _body(self, ro: Rollout, T: int):
        env = self.env
        self._refresh_heads()
        for t in range(T):
            # finished episodes start over (reference: the collector calls reset() on a fresh clone);
            # the kernels add the device clock: effective seed = seed + 0x9E3779B9 * (clock + t + 1)
            env.set_counters(t, t)  # coin / permutation draws: counter t + clock, eager or replayed
            if t == 0 or not self._fused_step:  # with the fused step the envs that finish in step t are reset by step t's own call
                env.reset_done(self.seed + 0x9E3779B9 * (t + 1))
            self._observe(ro, t)
            # masks() is all-true for a live env (clifford.rs:349-351), so sampling needs no mask
            if self._fused_step:  # first layer; then middle layer + head + draw + env.step + reset of the finished envs in one call
                # (the seed of step t + 1's reset_done; the device clock advances by T between collections, so the last step of a
                # collection resets with what the next collection's first reset_done would use)
                mid_head_sample_step(env, self._first_layer(), self._mid, self.policy.common.out_features, self._head, self.seed, t, ro.actions[t],
                                     ro.logp[t], ro.entropy[t], ro.values[t], rewards=ro.rewards[t], dones=ro.dones[t],
                                     reset_seed=self.seed + 0x9E3779B9 * (t + 2))
                continue
            self._forward_sample(ro, t)
            env.rollout(ro.actions[t : t + 1], rewards_out=ro.rewards[t : t + 1], dones_out=ro.dones[t : t + 1])
        # bootstrap value of the state after the last step (masked by `dones` where the episode ended)
        if env.env_kind == "pauli":
            last_v = None  # observing would draw a permutation; episodes are short, bootstrap with 0
        else:
            last_v = self._value_of_current_state()
            if ro.last_values is None:
                ro.last_values = torch.empty(env.batch, dtype=torch.float32, device=env.device)
            ro.last_values.copy_(last_v)
        gae(ro.rewards, ro.values, ro.dones, ro.last_values, self.gamma, self.gae_lambda, advantages=ro.advantages, returns=ro.returns)
        self.clock.add_(T)

    @torch.no_grad()
    def collect(self, T: int, out: Optional[Rollout] = None) -> Rollout:
        if not self.use_graph:
            ro = self._alloc(T) if out is None else out
            self._body(ro, T)
            return ro
        if self._graph is not None and self._graph_T == T:
            self._graph.replay()
            return self._graph_ro
        # first call (or a new T): one eager pass (allocates scratch, warms the GEMM handles, and IS
        # this call's rollout), then capture the same body for the calls that follow
        ro = self._alloc(T)
        self._body(ro, T)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):  # records, does not run: the env state and the clock are untouched
            self._body(ro, T)
        torch.cuda.synchronize()
        self._graph, self._graph_T, self._graph_ro = graph, T, ro
        return ro
