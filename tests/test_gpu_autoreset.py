"""qg_vec_reset_done: episodes that are over are re-scrambled on the device, live ones are not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import OracleEnv  # noqa: E402
from util import f32_bits, grid_gateset, line_gateset, rng_actions  # noqa: E402


@pytest.mark.parametrize("kind,n,inverts", [("clifford", 16, False), ("clifford", 5, True), ("linear_function", 8, False),
                                            ("linear_function", 12, True), ("permutation", 9, False), ("permutation", 25, True),
                                            ("clifford", 20, False), ("clifford", 18, True), ("linear_function", 40, False)])  # 64-bit rows
def test_reset_done_only_touches_finished_episodes(kind, n, inverts):
    from qiskit_gym_amd.vec import VecEnv

    side = int(round(n ** 0.5))
    gs = grid_gateset("permutation", side, side) if kind == "permutation" else line_gateset(kind, n)
    A, B, diff = len(gs), 333, 3
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=True, difficulty=diff, depth_slope=2, max_depth=128)
    gv = VecEnv(kind, n, gs, B, **cfg)
    envs = [OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(B)]
    seed0 = 11
    gv.reset(seed0)
    draws = rng_actions(seed0, B, diff, A)
    for e, o in enumerate(envs):
        o.reset_with(draws[:, e])
    rng = np.random.default_rng(4)
    n_resets = 0
    for t in range(30):
        # episodes end after depth_slope*difficulty = 6 steps (or on success): auto-reset them
        done = gv.done.cpu().numpy().astype(bool)
        want_done = np.array([o.is_final() for o in envs])
        np.testing.assert_array_equal(done, want_done)
        if done.any():
            seed = 1000 + t
            gv.reset_done(seed)
            d2 = rng_actions(seed, B, diff, A)
            for e in np.nonzero(done)[0]:
                envs[e].reset_with(d2[:, e])
                n_resets += 1
        acts = rng.integers(0, A, size=B)
        coins = rng.integers(0, 2, size=B)
        for o, a, c in zip(envs, acts, coins):
            o.step(int(a), int(c))
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32), torch.as_tensor(coins, device="cuda", dtype=torch.uint8))
        gv.sync()
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32), err_msg=f"t={t}")
        np.testing.assert_array_equal(gv.depth.cpu().numpy(), [o.depth() for o in envs])
    assert n_resets > B  # every env went through several episodes
    np.testing.assert_array_equal(gv.get_state("i64").cpu().numpy(), np.stack([o.get_state() for o in envs]))
    for e in (0, 7, B - 1):
        assert gv.solution(e) == envs[e].solution()


def test_pauli_reset_done_generates_fresh_targets_on_device():
    """PauliEnv episodes that are over get a freshly generated target (device-side generator);
    live episodes are untouched."""
    from qiskit_gym_amd.vec import VecEnv

    n, B = 6, 150
    gs = line_gateset("pauli", n)
    A = len(gs)
    cfg = dict(add_perms=False, track_solution=True, max_rotations=4, difficulty=9, pauli_diff_scale=3, depth_slope=1, max_depth=64)
    gv = VecEnv("pauli", n, gs, B, **cfg)
    envs = [OracleEnv("pauli", n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(B)]
    gv.reset(5)
    for e, o in enumerate(envs):
        o.pauli_reset_seeded(5, e)
    rng = np.random.default_rng(2)
    n_resets = 0
    for t in range(40):
        done = gv.done.cpu().numpy().astype(bool)
        np.testing.assert_array_equal(done, [o.is_final() for o in envs])
        if done.any():
            gv.reset_done(900 + t)
            for e in np.nonzero(done)[0]:
                envs[e].pauli_reset_seeded(900 + t, int(e))
                n_resets += 1
        acts = rng.integers(0, A, size=B)
        for o, a in zip(envs, acts):
            o.step(int(a))
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32))
        gv.sync()
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32), err_msg=f"t={t}")
    assert n_resets > B
    np.testing.assert_array_equal(gv.observe().cpu().numpy(), np.stack([o.dense_obs() for o in envs]))
    for e in (0, 11, B - 1):
        assert gv.solution(e) == envs[e].solution()
