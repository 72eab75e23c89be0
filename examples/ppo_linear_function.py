#!/usr/bin/env python3
"""End-to-end example: PPO on LinearFunctionGym with everything on one MI355X.

Collection (auto-reset, observation, policy forward, sampling, env.step, GAE) is the GPU-resident
`RolloutCollector`; the update below is a plain torch PPO step (clipped surrogate + value loss + entropy
bonus) with the reference's default hyper-parameters (rl/configs.py:134-144: gamma = lambda = 0.995,
clip_ratio 0.1, 10 epochs).  The trainers themselves are out of this repo's scope;
this script only shows that the collector's rollouts are what a learner needs.

    python examples/ppo_linear_function.py [--qubits 4] [--difficulty 5] [--iters 30]
    python examples/ppo_linear_function.py --env clifford --bf16      # CliffordGym, collection on the policy-layer kernels
    python examples/ppo_linear_function.py --env pauli --qubits 3 --bf16   # PauliGym: targets generated on the device, first layer from the packed observation words

With --bf16 the learner keeps f32 master weights and the collector a bf16 copy (refreshed, i.e. re-packed for
qg_vec_embed / qg_policy_mid_head_sample, before every collection): the forward pass and the draw of the collection
then run on libqgym's MFMA kernels straight from the envs' bit-packed state, the update is still plain torch.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
from qiskit_gym_amd.envs import CliffordGym, LinearFunctionGym, PauliGym


def train(qubits=4, difficulty=5, envs=4096, horizon=12, iters=30, epochs=4, minibatches=4, lr=3e-4, clip=0.1, vf_coef=0.5,
          ent_coef=0.01, seed=0, log=print, env_kind="linear_function", bf16=False, return_policy=False):
    torch.manual_seed(seed)
    edges = [(i, i + 1) for i in range(qubits - 1)] + [(i + 1, i) for i in range(qubits - 1)]
    cls = {"linear_function": LinearFunctionGym, "clifford": CliffordGym, "pauli": PauliGym}[env_kind]
    gym = cls.from_coupling_map(edges, difficulty=difficulty, add_inverts=False, add_perms=False)  # (PauliGym has no add_inverts: filtered out)
    env = gym.vec(batch=envs, track_solution=False)
    rows, cols = env.obs_shape_
    if bf16:  # the default policy shape: both policy-layer kernels apply (hidden 512 % 64 == 0, middle layer 256)
        policy = BasicPolicy(rows * cols, env.num_actions_).to(env.device)
        actor = BasicPolicy(rows * cols, env.num_actions_)
        actor.load_state_dict(policy.state_dict())
        col = RolloutCollector(env, actor, dtype=torch.bfloat16, seed=seed, store_obs="packed", use_bit_embedding=True, use_fused_head=True)
    else:
        policy = BasicPolicy(rows * cols, env.num_actions_, embedding_size=256, common=128)
        col = RolloutCollector(env, policy, dtype=torch.float32, seed=seed, store_obs="packed")
    opt = torch.optim.Adam(policy.parameters(), lr=lr)
    history = []
    for it in range(iters):
        t0 = time.perf_counter()
        if bf16:  # master weights -> the collector's bf16 copy, in place (the collector re-packs them when it starts)
            with torch.no_grad():
                for dst, src in zip(col.policy.parameters(), policy.parameters()):
                    dst.copy_(src)
        ro = col.collect(horizon)
        T, B = ro.actions.shape
        obs = ro.dense_obs(torch.float32).reshape(T * B, -1)
        act, old_logp = ro.actions.reshape(-1), ro.logp.reshape(-1)
        adv, ret = ro.advantages.reshape(-1), ro.returns.reshape(-1)
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        for _ in range(epochs):
            perm = torch.randperm(T * B, device=obs.device)
            for mb in perm.chunk(minibatches):
                logits, value = policy(obs[mb])
                logp_all = torch.log_softmax(logits, dim=-1)
                logp = logp_all.gather(1, act[mb].unsqueeze(1)).squeeze(1)
                ratio = torch.exp(logp - old_logp[mb])
                surr = torch.minimum(ratio * adv[mb], torch.clamp(ratio, 1 - clip, 1 + clip) * adv[mb])
                entropy = -(logp_all.exp() * logp_all).sum(-1).mean()
                loss = -surr.mean() + vf_coef * (value - ret[mb]).pow(2).mean() - ent_coef * entropy
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
        # fraction of finished episodes that ended in success (an episode ends by success or by running out of depth)
        ends = ro.dones.bool()
        solved = (ro.rewards > 0.5) & ends
        rate = float(solved.sum()) / max(1, int(ends.sum()))
        history.append(rate)
        log(f"iter {it:3d}: {int(ends.sum()):6d} episodes, solved {rate:6.1%}, mean reward/step {float(ro.rewards.mean()):+.4f}, "
            f"{T * B / (time.perf_counter() - t0):.2e} env-steps/s incl. update")
    env.sync()
    return (history, gym, policy) if return_policy else history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--qubits", type=int, default=4)
    ap.add_argument("--difficulty", type=int, default=5)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--env", default="linear_function", choices=["linear_function", "clifford", "pauli"])
    ap.add_argument("--bf16", action="store_true")
    a = ap.parse_args()
    train(qubits=a.qubits, difficulty=a.difficulty, iters=a.iters, envs=a.envs, env_kind=a.env, bf16=a.bf16)
