"""The GPU-resident collector's trajectories replayed on the CPU oracle: same observations,
rewards and episode boundaries, including the on-device auto-resets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import OracleEnv  # noqa: E402
from util import f32_bits, line_gateset, rng_actions  # noqa: E402


def test_collector_trajectories_replay_on_the_oracle():
    from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
    from qiskit_gym_amd.vec import VecEnv

    n, B, T, diff = 4, 192, 24, 2
    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    env = VecEnv("clifford", n, gs, B, **cfg)
    torch.manual_seed(0)
    pol = BasicPolicy(4 * n * n, A, embedding_size=64, common=32)
    col = RolloutCollector(env, pol, dtype=torch.float32, seed=77)
    ro = col.collect(T)
    torch.cuda.synchronize()
    env.sync()
    obs, acts = ro.obs.cpu().numpy(), ro.actions.cpu().numpy()
    rew, done = ro.rewards.cpu().numpy(), ro.dones.cpu().numpy()
    assert acts.min() >= 0 and acts.max() < A
    envs = [OracleEnv("clifford", n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(B)]
    n_episodes = 0
    for t in range(T):
        seed = (77 + 0x9E3779B9 * (t + 1)) & (2**64 - 1)
        draws = rng_actions(seed, B, diff, A)
        for e, o in enumerate(envs):
            if o.is_final():  # a fresh env is final (depth 1, solved): the first step resets everyone
                o.reset_with(draws[:, e])
                n_episodes += 1
            np.testing.assert_array_equal(o.dense_obs().reshape(-1), obs[t, e], err_msg=f"obs t={t} env={e}")
            o.step(int(acts[t, e]))
            assert o.reward_bits() == int(f32_bits(rew[t, e])), (t, e)
            assert int(o.is_final()) == int(done[t, e]), (t, e)
    assert n_episodes > B  # several episodes per env (depth_slope * difficulty = 4 steps)
    assert torch.isfinite(ro.logp).all() and torch.isfinite(ro.values).all()
