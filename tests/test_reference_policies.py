"""The policies the reference ships trained (examples/models/*.pt, converted to tests/golden/policies/*.npz by
tests/golden/make_golden.py) as known-answer data for the env.

A BasicPolicy trained against the reference's Rust env reads the flattened observation and answers with an action
index.  Run greedily it solves random targets only if observation layout, action order, gate semantics and the
solved test are the reference's: on the CPU oracle and on the HIP path it solves (nearly) every target, and on an
env with one of those things changed it does not.  This pins CliffordEnv / LinearFunctionEnv / PermutationEnv
stepping beyond the notebook transcripts (test_oracle_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import OracleEnv

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = {"clifford_3q_custom": "clifford", "lf_5_line": "linear_function", "perm_square_3x3": "permutation"}


def load(name):
    gs_all = json.load(open(os.path.join(HERE, "golden", "gatesets.json")))
    cfg = gs_all["model_" + name]["env"]
    gateset = [(g[0], tuple(g[1])) for g in cfg["gateset"]]
    z = np.load(os.path.join(HERE, "golden", "policies", name + ".npz"))
    w = {k: z[k].astype(np.float32) for k in z.files}
    return cfg, gateset, w


def greedy(w, obs):
    """twisterl BasicPolicy forward (Linear-ReLU-Linear-ReLU-Linear), argmax action; obs [..., obs_size]."""
    h = np.maximum(obs @ w["embeddings_weight"].T + w["embeddings_bias"], 0)
    h = np.maximum(h @ w["common_0_weight"].T + w["common_0_bias"], 0)
    return (h @ w["action_0_weight"].T + w["action_0_bias"]).argmax(axis=-1)


def solve_rate(kind, cfg, gateset, w, difficulty, episodes, seed, max_steps=96):
    rng = np.random.default_rng(seed)
    A, solved = len(gateset), 0
    for _ in range(episodes):
        env = OracleEnv(kind, cfg["num_qubits"], gateset, add_inverts=0, add_perms=0, track_solution=0, difficulty=difficulty,
                        depth_slope=cfg["depth_slope"], max_depth=cfg["max_depth"])
        env.reset_with(rng.integers(0, A, size=difficulty))
        t = 0
        while not env.success() and t < max_steps:
            env.step(int(greedy(w, env.dense_obs().reshape(-1).astype(np.float32))))
            t += 1
        solved += int(env.success())
    return solved / episodes


@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("difficulty", [4, 16, 32])
def test_reference_policy_solves_targets_on_the_oracle(name, difficulty):
    cfg, gateset, w = load(name)
    assert w["embeddings_weight"].shape[1] == OracleEnv(MODELS[name], cfg["num_qubits"], gateset).obs_shape()[0] ** 2
    assert w["action_0_weight"].shape[0] == len(gateset)
    rate = solve_rate(MODELS[name], cfg, gateset, w, difficulty, episodes=120, seed=difficulty)
    assert rate >= 0.97, f"{name}: the reference's policy solved only {rate:.2%} at difficulty {difficulty}"


def test_the_policies_notice_a_wrong_env():
    """Negative controls: the same policies on envs that differ from the reference in one respect."""
    cfg, gateset, w = load("lf_5_line")
    flipped = [(n, (q[1], q[0])) for n, q in gateset]  # CX control and target exchanged
    assert solve_rate("linear_function", cfg, flipped, w, 16, 60, 1) < 0.5
    cfg, gateset, w = load("clifford_3q_custom")
    hs = [({"H": "S", "S": "H"}.get(n, n), q) for n, q in gateset]  # H and S exchanged
    assert solve_rate("clifford", cfg, hs, w, 16, 60, 2) < 0.5
    rot = gateset[1:] + gateset[:1]  # action indices shifted by one
    assert solve_rate("clifford", cfg, rot, w, 16, 60, 3) < 0.5
    cfg, gateset, w = load("perm_square_3x3")
    assert solve_rate("permutation", cfg, gateset[::-1], w, 16, 60, 4) < 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_reference_policy_solves_targets_on_the_hip_path(name):
    """The same check through libqgym: 4096 envs scrambled on the device, the reference's policy in torch on the
    observation the library writes.  Solved envs are parked with an out-of-range action (a no-op for the state)."""
    import torch

    from qiskit_gym_amd.vec import VecEnv

    cfg, gateset, w = load(name)
    B, A = 4096, len(gateset)
    env = VecEnv(MODELS[name], cfg["num_qubits"], gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=24,
                 depth_slope=cfg["depth_slope"], max_depth=cfg["max_depth"])
    env.reset(7)
    tw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    solved = env.success.bool().clone()
    for _ in range(96):
        obs = env.observe_as(torch.float32)
        h = torch.relu(obs @ tw["embeddings_weight"].t() + tw["embeddings_bias"])
        h = torch.relu(h @ tw["common_0_weight"].t() + tw["common_0_bias"])
        act = (h @ tw["action_0_weight"].t() + tw["action_0_bias"]).argmax(dim=1).to(torch.int32)
        env.step(torch.where(solved, torch.full_like(act, A), act))
        solved |= env.success.bool()
    env.sync()
    rate = float(solved.float().mean())
    assert rate >= 0.97, f"{name}: {rate:.2%} of {B} targets solved"


@pytest.mark.gpu
def test_reference_clifford_policy_through_the_bit_consuming_first_layer():
    """CliffordEnv 3q is a TILE-layout env: the reference's first layer runs on qg_vec_embed (bf16), the rest in torch."""
    import torch

    from qiskit_gym_amd.collector import embed, pack_embedding
    from qiskit_gym_amd.vec import VecEnv

    cfg, gateset, w = load("clifford_3q_custom")
    B, A = 4096, len(gateset)
    env = VecEnv("clifford", 3, gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=24,
                 depth_slope=cfg["depth_slope"], max_depth=cfg["max_depth"])
    env.reset(11)
    tw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    packed = pack_embedding(env, tw["embeddings_weight"])
    solved = env.success.bool().clone()
    for _ in range(96):
        h = embed(env, packed, tw["embeddings_bias"], 512, relu=True).float()
        h = torch.relu(h @ tw["common_0_weight"].t() + tw["common_0_bias"])
        act = (h @ tw["action_0_weight"].t() + tw["action_0_bias"]).argmax(dim=1).to(torch.int32)
        env.step(torch.where(solved, torch.full_like(act, A), act))
        solved |= env.success.bool()
    env.sync()
    assert float(solved.float().mean()) >= 0.97


@pytest.mark.parametrize("name", ["clifford_3q_custom", "lf_5_line"])
def test_reference_value_heads_agree_with_the_reward_scheme(name):
    """Soft evidence for the reward constants (metrics.rs:135-146 with the default weights: 0.0101 per CX): the trained
    value head predicts the discounted return (gamma = 0.995) of the greedy rollout under the oracle's rewards to within
    0.02 for states one to three gates from solved -- it would sit ~0.006 higher per CX under the older scheme the
    notebook's printed rewards come from.  (The permutation checkpoint's value head does not
    track the current SWAP penalty of 0.0303 and is left out.)"""
    cfg, gateset, w = load(name)
    A = len(gateset)

    def value(o):
        h = np.maximum(o @ w["embeddings_weight"].T + w["embeddings_bias"], 0)
        h = np.maximum(h @ w["common_0_weight"].T + w["common_0_bias"], 0)
        return (h @ w["value_0_weight"].T + w["value_0_bias"]).item()

    for scramble in (1, 2, 3):
        rng = np.random.default_rng(scramble)
        vs, gs = [], []
        for _ in range(60):
            env = OracleEnv(MODELS[name], cfg["num_qubits"], gateset, add_inverts=0, add_perms=0, track_solution=0, difficulty=scramble)
            env.reset_with(rng.integers(0, A, size=scramble))
            if env.success():
                continue
            vs.append(value(env.dense_obs().reshape(-1).astype(np.float32)))
            ret, disc, t = 0.0, 1.0, 0
            while not env.success() and t < 20:
                env.step(int(greedy(w, env.dense_obs().reshape(-1).astype(np.float32))))
                ret += disc * env.reward()
                disc *= 0.995
                t += 1
            gs.append(ret)
        assert abs(np.mean(vs) - np.mean(gs)) < 0.02, (name, scramble, np.mean(vs), np.mean(gs))
