"""The GPU-resident collector's trajectories replayed on the CPU oracle: same observations,
rewards and episode boundaries, including the on-device auto-resets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import OracleEnv  # noqa: E402
from collect_ref import gae_f32  # noqa: E402
from util import f32_bits, line_gateset, rng_actions  # noqa: E402


@pytest.mark.parametrize("store_obs,dtype_name", [("dense", "float32"), ("packed", "float32"), ("packed", "bfloat16"), ("dense", "bfloat16")])
def test_collector_trajectories_replay_on_the_oracle(store_obs, dtype_name):
    from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
    from qiskit_gym_amd.vec import VecEnv

    n, B, T, diff = 4, 192, 24, 2
    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    env = VecEnv("clifford", n, gs, B, **cfg)
    torch.manual_seed(0)
    pol = BasicPolicy(4 * n * n, A, embedding_size=64, common=32)
    dtype = getattr(torch, dtype_name)
    col = RolloutCollector(env, pol, dtype=dtype, seed=77, gamma=0.99, gae_lambda=0.9, store_obs=store_obs, use_bit_embedding=True, use_fused_head=True)
    # in bf16 the first layer reads the env's bit-packed state directly (qg_vec_embed), no dense policy input exists
    assert (col._embed is not None) == (dtype == torch.bfloat16)
    tol = 1e-4 if dtype == torch.float32 else 6e-2  # bf16 activations: 2^-8 relative per layer
    ro = col.collect(T)
    torch.cuda.synchronize()
    env.sync()
    assert ro.obs_packed == (store_obs == "packed")
    obs, acts = ro.dense_obs(torch.int8).cpu().numpy(), ro.actions.cpu().numpy()
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
    assert col.steps_done == T
    assert torch.isfinite(ro.logp).all() and torch.isfinite(ro.values).all() and (ro.logp <= 0).all() and (ro.entropy >= 0).all()
    # the policy input the collector built is the observation; the sampled log-probs are the policy's
    logits, value = col.policy.float()(torch.from_numpy(obs[T - 1]).cuda().float())
    lsm = torch.log_softmax(logits.float(), dim=-1)
    torch.testing.assert_close(ro.logp[T - 1], lsm.gather(1, ro.actions[T - 1].unsqueeze(1)).squeeze(1), atol=tol, rtol=0)
    torch.testing.assert_close(ro.values[T - 1], value.float(), atol=tol, rtol=0)
    # GAE over the rollout, bootstrapped with the value of the state after the last step
    want_adv, want_ret = gae_f32(rew, ro.values.cpu().numpy(), done, ro.last_values.cpu().numpy(), 0.99, 0.9)
    np.testing.assert_array_equal(f32_bits(ro.advantages.cpu().numpy()), f32_bits(want_adv))
    np.testing.assert_array_equal(f32_bits(ro.returns.cpu().numpy()), f32_bits(want_ret))


def test_collector_runs_pauli_and_generic_policies():
    """PauliGym (dense observation only, device-side target generator) and a policy that is not BasicPolicy."""
    from qiskit_gym_amd.collector import RolloutCollector
    from qiskit_gym_amd.vec import VecEnv

    n, B, T = 5, 96, 12
    gs = line_gateset("pauli", n)
    env = VecEnv("pauli", n, gs, B, add_perms=False, track_solution=False, max_rotations=3, difficulty=4, depth_slope=1)
    rows, cols = env.obs_shape_

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Linear(rows * cols, 48)
            self.pi = torch.nn.Linear(48, len(gs))
            self.v = torch.nn.Linear(48, 1)

        def forward(self, x):
            h = torch.tanh(self.body(x))
            return self.pi(h), self.v(h).squeeze(-1)

    torch.manual_seed(1)
    col = RolloutCollector(env, Tiny(), dtype=torch.float32, seed=3)
    ro = col.collect(T)
    torch.cuda.synchronize()
    env.sync()
    assert ro.obs.shape == (T, B, rows * cols) and ro.dones.sum() > 0
    assert torch.isfinite(ro.advantages).all() and ro.last_values is None
    # the packed form of the PauliGym observation (one 64-bit word per row) stores the same observations
    env2 = VecEnv("pauli", n, gs, B, add_perms=False, track_solution=False, max_rotations=3, difficulty=4, depth_slope=1)
    torch.manual_seed(1)
    col2 = RolloutCollector(env2, Tiny(), dtype=torch.float32, seed=3, store_obs="packed")
    ro2 = col2.collect(T)
    torch.cuda.synchronize()
    env2.sync()
    assert ro2.obs_packed and ro2.obs.shape == (T, B, rows) and ro2.obs.dtype == torch.int64
    assert torch.equal(ro2.dense_obs(torch.int8), ro.obs) and torch.equal(ro2.actions, ro.actions) and torch.equal(ro2.rewards, ro.rewards)



@pytest.mark.parametrize("kind,n,kw", [("clifford", 4, dict(add_inverts=True)), ("linear_function", 6, dict(add_inverts=False)),
                                       ("pauli", 5, dict(max_rotations=3, depth_slope=1))])
def test_graph_replays_collect_the_same_rollouts_as_eager_calls(kind, n, kw):
    """use_graph=True: the first call runs eagerly and captures, later calls replay one hipGraph.  The
    device clock makes every replay draw fresh resets / actions / inversion coins, and they are the
    very draws the eager collector makes at the same step numbers."""
    from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
    from qiskit_gym_amd.vec import VecEnv

    B, T, calls = 160, 6, 4
    gs = line_gateset(kind, n)
    cfg = dict(add_perms=False, track_solution=False, difficulty=3, **kw)
    rollouts = {}
    for mode in (False, True):
        env = VecEnv(kind, n, gs, B, **cfg)
        torch.manual_seed(3)
        pol = BasicPolicy(env.obs_shape_[0] * env.obs_shape_[1], len(gs), embedding_size=64, common=32)
        col = RolloutCollector(env, pol, dtype=torch.float32, seed=21, use_graph=mode)
        got = []
        for _ in range(calls):
            ro = col.collect(T)
            torch.cuda.synchronize()
            got.append({k: getattr(ro, k).clone() for k in ("obs", "actions", "logp", "values", "rewards", "dones", "advantages")})
        env.sync()
        assert col.steps_done == calls * T
        rollouts[mode] = got
    for k in range(calls):
        for name, eager in rollouts[False][k].items():
            if eager.dtype.is_floating_point:  # the GEMMs may pick different kernels under capture
                torch.testing.assert_close(rollouts[True][k][name], eager, atol=1e-4, rtol=1e-4, msg=lambda m: f"call {k} {name}: {m}")
            else:
                assert torch.equal(rollouts[True][k][name], eager), (k, name)
    # successive rollouts differ (the clock advanced)
    assert not torch.equal(rollouts[True][1]["actions"], rollouts[True][2]["actions"])
    assert sum(int(r["dones"].sum()) for r in rollouts[True]) > B


@pytest.mark.parametrize("kind,n,inverts,B", [("clifford", 6, False, 256), ("linear_function", 12, True, 256), ("linear_function", 12, True, 9000)])
def test_graph_replay_with_the_bit_consuming_first_layer(kind, n, inverts, B):
    """bf16 policy on an env whose rows are resident uint32 words (TILE layout; LinearFunctionEnv with add_inverts, the reference's
    default: the dual layout): the captured collection launches qg_vec_embed instead of observe + GEMM, and a replay after an
    in-place parameter update uses the repacked weights."""
    from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
    from qiskit_gym_amd.vec import VecEnv

    T = 5
    gs = line_gateset(kind, n)
    env = VecEnv(kind, n, gs, B, add_inverts=inverts, add_perms=False, track_solution=False, difficulty=3)
    torch.manual_seed(5)
    pol = BasicPolicy(env.obs_shape_[0] * env.obs_shape_[1], len(gs), embedding_size=128, common=64)
    col = RolloutCollector(env, pol, dtype=torch.bfloat16, seed=9, store_obs="packed", use_graph=True, use_bit_embedding=True, use_fused_head=True)
    assert col._embed is not None
    for call in range(3):
        ro = col.collect(T)
        torch.cuda.synchronize()
        obs = ro.dense_obs(torch.float32)[T - 1]
        logits, value = col.policy(obs.to(torch.bfloat16))
        lsm = torch.log_softmax(logits.float(), dim=-1)
        torch.testing.assert_close(ro.logp[T - 1], lsm.gather(1, ro.actions[T - 1].unsqueeze(1)).squeeze(1), atol=6e-2, rtol=0)
        torch.testing.assert_close(ro.values[T - 1], value.float(), atol=6e-2, rtol=0)
        with torch.no_grad():  # an optimiser step, in place: the next replay must see it (heads refreshed, first layer repacked)
            for prm in col.policy.parameters():
                prm.mul_(0.5)
    env.sync()
    assert col.steps_done == 3 * T


@pytest.mark.parametrize("kind,n,kw,use_graph", [("pauli", 6, dict(max_rotations=4, depth_slope=1), False), ("pauli", 6, dict(max_rotations=4, depth_slope=1), True),
                                                 ("clifford", 18, dict(add_inverts=False), True)])
def test_first_layer_from_the_packed_observation_words(kind, n, kw, use_graph):
    """bf16 BasicPolicy on an env whose packed observation is one 64-bit word per row (PauliGym, CliffordGym N > 16): the collector stores
    the words and feeds them straight to qg_policy_embed_words -- no dense observation exists.  log-probs / values must match the torch
    policy on the re-expanded stored observation, also after an in-place parameter update (repacked weights)."""
    from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
    from qiskit_gym_amd.vec import VecEnv

    B, T = 300, 5
    gs = line_gateset(kind, n)
    env = VecEnv(kind, n, gs, B, add_perms=False, track_solution=False, difficulty=3, **kw)
    rows, cols = env.obs_shape_
    torch.manual_seed(5)
    pol = BasicPolicy(rows * cols, len(gs), embedding_size=128, common=64)
    col = RolloutCollector(env, pol, dtype=torch.bfloat16, seed=9, store_obs="packed", use_graph=use_graph, use_bit_embedding=True, use_fused_head=True)
    assert col._embed is None and col._embed_words is not None
    for call in range(3):
        ro = col.collect(T)
        torch.cuda.synchronize()
        assert ro.obs.dtype == torch.int64 and ro.obs.shape == (T, B, rows)
        for t in (0, T - 1):
            obs = ro.dense_obs(torch.float32)[t]
            logits, value = col.policy(obs.to(torch.bfloat16))
            lsm = torch.log_softmax(logits.float(), dim=-1)
            torch.testing.assert_close(ro.logp[t], lsm.gather(1, ro.actions[t].unsqueeze(1)).squeeze(1), atol=6e-2, rtol=0)
            torch.testing.assert_close(ro.values[t], value.float(), atol=6e-2, rtol=0)
        with torch.no_grad():
            for prm in col.policy.parameters():
                prm.mul_(0.5)
    env.sync()
    assert col.steps_done == 3 * T


@pytest.mark.parametrize("kind,n,use_graph,B,inverts", [("clifford", 5, False, 1000, False), ("clifford", 16, True, 1000, False), ("linear_function", 12, False, 1000, False),
                                                         ("clifford", 5, True, 9000, False), ("linear_function", 20, False, 8193, False),  # > 8 192 envs: the reset is its own launch
                                                         # add_inverts, the reference's default: the two-lanes-per-env step inside the small kernel; separate launches beyond
                                                         ("clifford", 16, True, 1000, True), ("clifford", 7, False, 333, True), ("clifford", 12, True, 8200, True)])
def test_sampling_kernel_that_also_steps_the_env_equals_the_separate_launches(kind, n, use_graph, B, inverts):
    """qg_vec_mid_head_sample_step_reset: middle layer + head + draw + Env::step + the reset of the envs that finished, in one call (one
    launch for small batches).  Same trajectories, bit for bit, as the sampling kernel followed by qg_vec_step and a reset_done per step
    -- and the trajectories replay on the oracle (auto-resets included)."""
    from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
    from qiskit_gym_amd.vec import VecEnv

    T, diff = 12, 2
    gs = line_gateset(kind, n)
    A = len(gs)
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=inverts, difficulty=diff)  # with the solution log in the default configuration
    obs_size = 4 * n * n if kind == "clifford" else n * n
    runs = []
    for fused in (True, False):
        env = VecEnv(kind, n, gs, B, **cfg)
        torch.manual_seed(3)
        pol = BasicPolicy(obs_size, A, embedding_size=128, common=256)  # the fused tail needs the reference's 256 middle features
        col = RolloutCollector(env, pol, dtype=torch.bfloat16, seed=5, store_obs="packed", use_graph=use_graph, use_bit_embedding=True,
                               use_fused_head=True, use_fused_step=fused)
        assert col._mid is not None and col._fused_step == fused
        got = []
        for call in range(2):
            ro = col.collect(T)
            torch.cuda.synchronize()
            got.append({k: getattr(ro, k).clone() for k in ("obs", "actions", "logp", "values", "rewards", "dones", "advantages")})
        if not fused:  # the fused call has already reset the envs that finished in the last step: what the next collection's first
            env.reset_done(5 + 0x9E3779B9)  # reset_done (seed + phi * 1, on the clock both envs now share) is about to do here
        env.sync()
        runs.append((got, env.get_state("packed").clone(), env.depth.clone()))
    for call in range(2):
        for k in runs[0][0][call]:
            assert torch.equal(runs[0][0][call][k], runs[1][0][call][k]), (call, k)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    # oracle replay of the fused run's first collection
    ro = runs[0][0][0]
    acts, rew, done = ro["actions"].cpu().numpy(), ro["rewards"].cpu().numpy(), ro["dones"].cpu().numpy()
    stride = 7 if B <= 1000 else 61
    envs = [OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(0, B, stride)]
    for t in range(T):
        draws = rng_actions((5 + 0x9E3779B9 * (t + 1)) & (2**64 - 1), B, diff, A)
        # add_inverts: the handle's counter RNG throws the coin of env e at step t (qgym_api.cpp: coin_seed; device_common: rng_draw(.., env, step + clock) >> 63)
        from util import rng_draw
        coins = (rng_draw(0x5EED0000C01F ^ 0x636F696E, np.arange(0, B, stride, dtype=np.uint64), t) >> np.uint64(63)).astype(np.int64)
        for i, o in enumerate(envs):
            e = stride * i
            if o.is_final():
                o.reset_with(draws[:, e])
            o.step(int(acts[t, e]), int(coins[i]) if inverts else 0)
            assert o.reward_bits() == int(f32_bits(rew[t, e])) and int(o.is_final()) == int(done[t, e]), (t, e)
