"""Batched synthesis: many targets x many policy-guided searches as ONE env batch on the GPU.

The inference-side caller of the hot path.  The reference's `RLSynthesis.synth(input)`
(src/qiskit_gym/rl/synthesis.py:111-126) is `env.get_state(input)` -> twisterl's
`algorithm.solve(state, deterministic, num_searches, ...)` -> `env.build_circuit_from_solution`: per
target, `num_searches` episodes that clone the scalar env, `set_state` it and step it under the
policy until `is_final`, one target at a time on the CPU.  Here every (target, search) pair is one env
of a `VecEnv` batch: `set_state` once, then observe -> policy -> draw -> `env.step` for all of them per
launch, and the successful episode with the highest return (the env's own metrics-weighted reward,
metrics.rs:135-146) wins per target.  Same seed => same draws => same circuits.

The search batch runs without inversions and observation permutations (both are training-time
augmentations: clifford.rs:262-270, pauli.rs:653-665), so a solution is the winner's action sequence;
PauliGym solutions also carry the rotations each gate released (pauli.rs:612-626), which are
recovered by replaying the winners on a `track_solution` batch.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .collector import BasicPolicy, embed, embed_words, mid_head_sample, pack_embed_words, pack_embedding, pack_head, pack_mid, sample_actions
from .envs.gyms import ROTATION_MARKER
from .vec import VecEnv


def policy_from_reference_state_dict(sd: Dict[str, "np.ndarray | torch.Tensor"]) -> BasicPolicy:
    """A `BasicPolicy` holding a reference checkpoint (`examples/models/*.pt`: embeddings / common.0 / action.0 /
    value.0, weight + bias; keys with '.' or '_' separators)."""
    t = {k.replace(".", "_"): torch.as_tensor(np.asarray(v, dtype=np.float32) if not isinstance(v, torch.Tensor) else v).float() for k, v in sd.items()}
    hidden, obs_size = t["embeddings_weight"].shape
    pol = BasicPolicy(obs_size, t["action_0_weight"].shape[0], embedding_size=hidden, common=t["common_0_weight"].shape[0])
    with torch.no_grad():
        for mod, key in ((pol.embeddings, "embeddings"), (pol.common, "common_0"), (pol.policy_head, "action_0"), (pol.value_head, "value_0")):
            mod.weight.copy_(t[key + "_weight"])
            mod.bias.copy_(t[key + "_bias"])
    return pol


class BatchedSynthesis:
    """`env`: one of the *Gym front ends (its configuration and gateset are used); `policy`: a module mapping the flat
    observation [B, rows*cols] to (logits [B, num_actions], value [B])."""

    def __init__(self, env, policy: torch.nn.Module, dtype: torch.dtype = torch.float32, seed: int = 0, device=None):
        self.env = env
        self.dtype = dtype
        self.seed = int(seed)
        self.device = device
        self._policy = policy
        self._vecs: Dict[tuple, VecEnv] = {}
        self._packed = None  # (vec, packed first layer, its f32 bias, packed middle layer, packed head): the policy-layer kernels' operands
        self.last_stats: dict = {}

    def _kernels(self, vec: VecEnv):
        """Operands of the two policy-layer kernels (qg_vec_embed, qg_policy_mid_head_sample: bf16 products, f32 accumulation) when the
        policy has the default shape and the env a TILE layout or 64-bit observation words; None otherwise (the torch forward is used)."""
        if self._packed is not None and self._packed[0] is vec:
            return self._packed
        pol = self._policy
        if not isinstance(pol, BasicPolicy):
            return None
        try:
            w, b, A = pol.fused_heads()
            try:
                first = pack_embedding(vec, pol.embeddings.weight)  # TILE layout: the first layer reads the resident state
                words = False
            except (ValueError, _lib.QGymError):
                if vec.packed_word_bytes != 8 or vec.packed_words_per_env != vec.obs_shape_[0]:
                    raise
                first = pack_embed_words(pol.embeddings.weight, *vec.obs_shape_)  # 64-bit row words (PauliEnv, wide CliffordEnv): qg_policy_embed_words
                words = True
            self._packed = (vec, first, pol.embeddings.bias.detach().float().contiguous(), pack_mid(pol.common.weight, pol.common.bias),
                            pack_head(w, b, A, A, after_mid=True), words)
        except (ValueError, _lib.QGymError):
            self._packed = None
        return self._packed

    def _vec(self, batch: int, track_solution: bool) -> VecEnv:
        key = (batch, track_solution)
        if key not in self._vecs:
            for k in [k for k in self._vecs if k[1] == track_solution]:  # one batch size at a time: the handles own device memory
                self._vecs.pop(k).close()
            self._vecs[key] = self.env.vec(batch, device=self.device, add_inverts=False, add_perms=False, track_solution=track_solution)
            self._policy = self._policy.to(device=self._vecs[key].device, dtype=self.dtype)
        return self._vecs[key]

    def _load(self, vec: VecEnv, states: Sequence[Sequence[int]], repeat: int):
        if vec.env_kind == "pauli":
            n = vec.num_qubits
            tabs, labels = [], []
            for s in states:  # the set_state wire format (envs/synthesis.py:451-461): [rot_count, tableau..., len, chars, ...]
                s = list(s)
                tabs.append(np.asarray(s[1 : 1 + 4 * n * n], dtype=np.uint8))
                pos, rots = 1 + 4 * n * n, []
                for _ in range(int(s[0])):
                    ln = int(s[pos])
                    rots.append("".join(chr(c) for c in s[pos + 1 : pos + 1 + ln]))
                    pos += 1 + ln
                labels.append(rots)
            vec.pauli_reset_from(np.repeat(np.stack(tabs), repeat, axis=0), [l for l in labels for _ in range(repeat)])
        else:
            vec.set_state(np.repeat(np.asarray(states, dtype=np.int64), repeat, axis=0), fmt="i64")

    @torch.no_grad()
    def solve(self, states: Sequence[Sequence[int]], deterministic: bool = False, num_searches: int = 100, fast: Optional[bool] = None) -> List[Optional[List[int]]]:
        """One entry per target: `Env::solution()` of the best successful search, or None (rl/synthesis.py:121-126).
        fast: run the sampled searches' forward pass and draw on the policy-layer kernels (bf16 products; default: when they apply and the
        batch has at least 4 096 envs); solutions are valid either way -- the env decides what solves a target, the policy only proposes."""
        M = len(states)
        if M == 0:
            return []
        S = 1 if deterministic else max(1, int(num_searches))  # greedy episodes are all alike
        vec = self._vec(M * S, False)
        B, A, dev = vec.batch, vec.num_actions(), vec.device
        self._load(vec, states, S)
        T = int(vec._cfg.max_depth)
        actions = torch.empty((T, B), dtype=torch.int32, device=dev)
        finished = vec.success.bool().clone()  # a target that is already solved needs no gates
        solved_at = torch.where(finished, 0, -1).to(torch.int32)
        ret = torch.zeros(B, dtype=torch.float32, device=dev)
        parked = torch.full((B,), A, dtype=torch.int32, device=dev)  # out of range: no gate (clifford.rs:324)
        steps = 0
        kern = None
        if not deterministic and fast is not False and (fast or B >= 4096):
            kern = self._kernels(vec)
            if fast and kern is None:
                raise ValueError("fast=True needs a BasicPolicy of the default shape and an env whose state or packed observation the first-layer kernels read")
        if kern is not None:
            pol = self._policy
            h1 = torch.empty((B, pol.embeddings.out_features), dtype=torch.bfloat16, device=dev)
            act = torch.empty(B, dtype=torch.int32, device=dev)
            scratch = torch.empty((3, B), dtype=torch.float32, device=dev)
            words_buf = torch.empty((B, vec.packed_words_per_env), dtype=torch.int64, device=dev) if kern[5] else None
        self.last_stats = {"kernels": kern is not None}
        for t in range(T):
            if kern is not None:
                if kern[5]:
                    embed_words(vec.observe_packed(out=words_buf), vec.obs_shape_[1], kern[1], kern[2], h1.shape[1], relu=True, out=h1)
                else:
                    embed(vec, kern[1], kern[2], h1.shape[1], relu=True, out=h1)
                mid_head_sample(h1, kern[3], pol.common.out_features, kern[4], A, self.seed, t, actions=act, logp=scratch[0], entropy=scratch[1], values=scratch[2])
            else:
                x = vec.observe_as(self.dtype)
                logits = self._policy(x)[0]
                if deterministic:
                    act = logits.argmax(dim=1).to(torch.int32)
                else:
                    act = sample_actions(logits.contiguous(), self.seed, t)[0].to(torch.int32)
            actions[t] = torch.where(finished, parked, act)
            vec.step(actions[t])
            live = ~finished
            ret += torch.where(live, vec.reward, torch.zeros_like(ret))
            solved_at = torch.where(live & vec.success.bool(), t + 1, solved_at)
            finished |= vec.done.bool()
            steps = t + 1
            if t % 8 == 7 and bool(finished.all()):
                break
        vec.sync()
        ok = (solved_at >= 0).view(M, S)
        score = torch.where(ok, ret.view(M, S), torch.full((M, S), -float("inf"), device=dev))
        best = score.argmax(dim=1)
        idx = torch.arange(M, device=dev) * S + best
        lengths = solved_at[idx].cpu().numpy()
        found = ok.any(dim=1).cpu().numpy()
        win = actions[:steps, idx].t().contiguous()  # [M, steps]
        self.last_stats.update({"targets": M, "searches": S, "steps": steps, "solved": int(found.sum()),
                           "searches_solved": float(ok.float().mean()), "mean_gates": float(lengths[found].mean()) if found.any() else 0.0})
        if vec.env_kind != "pauli":
            w = win.cpu().numpy()
            return [w[m, : lengths[m]].tolist() if found[m] else None for m in range(M)]
        # PauliEnv: replay the winners with the solution log on; padding entries (no gate, no marker) are dropped
        rep = self._vec(M, True)
        self._load(rep, states, 1)
        keep = torch.arange(steps, device=dev).view(1, -1) < solved_at[idx].view(-1, 1)
        rep.rollout(torch.where(keep, win, torch.full_like(win, A)).t().contiguous())
        rep.sync()
        out: List[Optional[List[int]]] = []
        for m in range(M):
            out.append([v for v in rep.solution(m) if v >= ROTATION_MARKER or v < A] if found[m] else None)
        return out

    def synth(self, inputs, deterministic: bool = False, num_searches: int = 100):
        """`RLSynthesis.synth` over a list of inputs: circuits (needs qiskit) or None where no search succeeded."""
        sols = self.solve([self.env.get_state(x) for x in inputs], deterministic, num_searches)
        return [self.env.build_circuit_from_solution(s, x) if s is not None else None for s, x in zip(sols, inputs)]

    def gate_lists(self, solutions):
        """Solutions as `(name, qubits)` lists (rotation markers of PauliGym solutions are skipped)."""
        gs = self.env.config["gateset"]
        return [None if s is None else [(gs[a][0], tuple(gs[a][1])) for a in s if a < ROTATION_MARKER] for s in solutions]
