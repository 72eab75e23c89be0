"""GPU parity for PauliEnv (PauliNetworkGym): tableau + rotation tracking vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import OracleEnv  # noqa: E402
from util import f32_bits, line_gateset  # noqa: E402


def random_tableau(rng, n, steps, pairs):
    """random_clifford_tableau (reference rust/src/envs/pauli.rs:220-271): 70/15/15 CX/H/S mix."""
    d = 2 * n
    t = np.eye(d, dtype=np.uint8)
    for _ in range(steps):
        r = rng.random()
        if r > 0.3:
            q0, q1 = pairs[rng.integers(len(pairs))]
            t[q1] ^= t[q0]
            t[n + q0] ^= t[n + q1]
        elif r > 0.15:
            q = rng.integers(n)
            t[[q, n + q]] = t[[n + q, q]]
        else:
            q = rng.integers(n)
            t[n + q] ^= t[q]
    return t


def random_labels(rng, n, count, max_weight=4):
    out = []
    for _ in range(count):
        w = int(rng.integers(1, max_weight + 1))
        qs = rng.choice(n, size=min(w, n), replace=False)
        s = ["I"] * n
        for q in qs:
            s[n - 1 - q] = "XYZ"[rng.integers(3)]
        out.append("".join(s))
    return out


@pytest.mark.parametrize("n,batch,max_rot,final_layers", [(4, 96, 5, None), (20, 64, 5, None), (3, 130, 2, 6), (6, 50, 8, 12), (5, 70, 20, 28), (20, 64, 24, 30)])
def test_pauli_step_parity(n, batch, max_rot, final_layers):
    from qiskit_gym_amd.vec import VecEnv

    gateset = line_gateset("pauli", n)
    A = len(gateset)
    pairs = [g[1] for g in gateset if g[0] == "CX"]
    cfg = dict(add_perms=False, track_solution=True, max_rotations=max_rot, max_depth=64, difficulty=9)
    if final_layers is not None:
        cfg["final_pauli_layers"] = final_layers
    rmax = final_layers if final_layers is not None else max_rot + 2
    rng = np.random.default_rng(100 + n)
    gv = VecEnv("pauli", n, gateset, batch, **cfg)
    envs = [OracleEnv("pauli", n, gateset, **{k: int(v) for k, v in cfg.items()}) for _ in range(batch)]
    tabs, labs = [], []
    for e in range(batch):
        t = random_tableau(rng, n, int(rng.integers(0, 4 * n)), pairs)
        l = random_labels(rng, n, int(rng.integers(0, rmax + 1)), max_weight=min(4, n))
        if e == 0:
            t, l = np.eye(2 * n, dtype=np.uint8), []  # solved from the start
        if e == 1:
            l = ["I" * (n - 1) + "Z"]  # trivial rotation: removed by the initial clean
            t = np.eye(2 * n, dtype=np.uint8)
        envs[e].pauli_reset_from(t, l)
        tabs.append(t)
        labs.append(l)
    gv.pauli_reset_from(np.stack(tabs), labs)
    gv.sync()

    def compare(label):
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32), err_msg=f"reward {label}")
        np.testing.assert_array_equal(gv.success.cpu().numpy(), [int(o.success()) for o in envs], err_msg=f"success {label}")
        np.testing.assert_array_equal(gv.done.cpu().numpy(), [int(o.is_final()) for o in envs], err_msg=f"final {label}")
        np.testing.assert_array_equal(gv.depth.cpu().numpy(), [o.depth() for o in envs], err_msg=f"depth {label}")

    def compare_obs(label):
        got = gv.observe().cpu().numpy()
        want = np.stack([o.dense_obs() for o in envs])
        np.testing.assert_array_equal(got, want, err_msg=f"obs {label}")

    compare("after reset")
    compare_obs("after reset")
    assert gv.success.cpu().numpy()[0] == 1 and gv.success.cpu().numpy()[1] == 1
    for t in range(48):
        acts = rng.integers(0, A, size=batch)
        if t % 9 == 8:
            acts[::6] = A + 1
        for o, a in zip(envs, acts):
            o.step(int(a))
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int64))
        gv.sync()
        compare(f"t={t}")
        if t % 8 == 7:
            compare_obs(f"t={t}")
    compare_obs("end")
    np.testing.assert_array_equal(gv.get_state("i64").cpu().numpy(), np.stack([o.get_state() for o in envs]))
    for e in range(0, batch, 7):
        assert gv.solution(e) == envs[e].solution(), e


def test_pauli_set_state_wire_format_and_fused_rollout():
    from qiskit_gym_amd.vec import VecEnv

    n, batch = 5, 40
    gateset = line_gateset("pauli", n)
    A = len(gateset)
    pairs = [g[1] for g in gateset if g[0] == "CX"]
    cfg = dict(add_perms=False, track_solution=False, max_rotations=3)
    rng = np.random.default_rng(8)
    gv = VecEnv("pauli", n, gateset, batch, **cfg)
    envs = [OracleEnv("pauli", n, gateset, **{k: int(v) for k, v in cfg.items()}) for _ in range(batch)]
    stride = 1 + 4 * n * n + 5 * (n + 1)
    states = np.zeros((batch, stride), dtype=np.int64)
    for e in range(batch):
        t = random_tableau(rng, n, 12, pairs)
        labs = random_labels(rng, n, int(rng.integers(0, 6)), 3)  # more than max_rotations get dropped (pauli.rs:538)
        rec = [len(labs)] + (t.astype(np.int64) * 3).reshape(-1).tolist()
        for l in labs:
            rec += [len(l)] + [ord(c) for c in l]
        states[e, : len(rec)] = rec
        envs[e].set_state(states[e].tolist())
    gv.set_state(states, "i64")
    gv.sync()
    assert int(gv.depth[0]) == 128
    T = 24
    acts = rng.integers(0, A, size=(T, batch))
    rew = torch.zeros((T, batch), dtype=torch.float32, device="cuda")
    fin = torch.zeros((T, batch), dtype=torch.uint8, device="cuda")
    gv.rollout(torch.as_tensor(acts, device="cuda", dtype=torch.int32), fused=True, rewards_out=rew, dones_out=fin)
    gv.sync()
    want_r = np.zeros((T, batch), np.uint32)
    want_f = np.zeros((T, batch), np.uint8)
    for t in range(T):
        for e, o in enumerate(envs):
            o.step(int(acts[t, e]))
            want_r[t, e] = o.reward_bits()
            want_f[t, e] = o.is_final()
    np.testing.assert_array_equal(f32_bits(rew.cpu().numpy()), want_r)
    np.testing.assert_array_equal(fin.cpu().numpy(), want_f)
    np.testing.assert_array_equal(gv.observe().cpu().numpy(), np.stack([o.dense_obs() for o in envs]))


@pytest.mark.parametrize("n,batch,max_rot,act_dtype", [(4, 130, 5, "int32"), (20, 64, 5, "int64"), (24, 70, 6, "int32"), (20, 200, 12, "int32")])
def test_pauli_fused_rollout_matches_the_oracle_step_by_step(n, batch, max_rot, act_dtype):
    """T steps in one launch (compact layout: the tile lives in LDS for the rollout; max_rot 12: the dense register kernel): per-step
    rewards and episode ends, the final observation / state, and a one-step launch afterwards (the bookkeeping the fused kernel
    wrote back must be what the one-step kernel expects), with out-of-range actions in the mix and a ragged last tile."""
    from qiskit_gym_amd.vec import VecEnv

    gateset = line_gateset("pauli", n)
    A = len(gateset)
    pairs = [g[1] for g in gateset if g[0] == "CX"]
    cfg = dict(add_perms=False, track_solution=False, max_rotations=max_rot, max_depth=40, difficulty=9)
    rng = np.random.default_rng(300 + n + max_rot)
    gv = VecEnv("pauli", n, gateset, batch, **cfg)
    envs = [OracleEnv("pauli", n, gateset, **{k: int(v) for k, v in cfg.items()}) for _ in range(batch)]
    tabs, labs = [], []
    for e in range(batch):
        # short scrambles and light rotations: some episodes finish (and keep being stepped) inside the rollout
        t = random_tableau(rng, n, int(rng.integers(0, 6)), pairs)
        l = random_labels(rng, n, int(rng.integers(0, max_rot + 3)), max_weight=min(3, n))
        envs[e].pauli_reset_from(t, l)
        tabs.append(t)
        labs.append(l)
    gv.pauli_reset_from(np.stack(tabs), labs)
    T = 48  # runs past max_depth: depth saturates at 0
    acts = rng.integers(0, A, size=(T, batch))
    acts[5, ::3] = A
    acts[17, 1::4] = -1
    acts[40:, ::2] = rng.integers(0, min(A, 2 * n), size=acts[40:, ::2].shape)  # one-qubit gates
    rew = torch.zeros((T, batch), dtype=torch.float32, device="cuda")
    fin = torch.zeros((T, batch), dtype=torch.uint8, device="cuda")
    gv.rollout(torch.as_tensor(acts, device="cuda", dtype=getattr(torch, act_dtype)), fused=True, rewards_out=rew, dones_out=fin)
    gv.sync()
    want_r = np.zeros((T, batch), np.uint32)
    want_f = np.zeros((T, batch), np.uint8)
    for t in range(T):
        for e, o in enumerate(envs):
            o.step(int(acts[t, e]))
            want_r[t, e] = o.reward_bits()
            want_f[t, e] = o.is_final()
    np.testing.assert_array_equal(f32_bits(rew.cpu().numpy()), want_r)
    np.testing.assert_array_equal(fin.cpu().numpy(), want_f)
    np.testing.assert_array_equal(gv.success.cpu().numpy(), [int(o.success()) for o in envs])
    np.testing.assert_array_equal(gv.depth.cpu().numpy(), [o.depth() for o in envs])
    np.testing.assert_array_equal(gv.observe().cpu().numpy(), np.stack([o.dense_obs() for o in envs]))
    np.testing.assert_array_equal(gv.get_state("i64").cpu().numpy(), np.stack([o.get_state() for o in envs]))
    for k in range(6):  # one-step launches on what the fused kernel left behind
        last = rng.integers(0, A, size=batch)
        for o, a in zip(envs, last):
            o.step(int(a))
        gv.step(torch.as_tensor(last, device="cuda", dtype=torch.int32))
    gv.sync()
    np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32))
    np.testing.assert_array_equal(gv.observe().cpu().numpy(), np.stack([o.dense_obs() for o in envs]))


def test_pauli_add_perms_observe_and_action_unpermute():
    """add_perms=True: observe() permutes qubits by a drawn coupling-map automorphism and the next
    step() un-permutes the action with it (reference pauli.rs:594-599, 653-665)."""
    from qiskit_gym_amd.vec import VecEnv

    n, batch = 4, 64
    gateset = line_gateset("pauli", n)
    A = len(gateset)
    pairs = [g[1] for g in gateset if g[0] == "CX"]
    cfg = dict(add_perms=True, track_solution=True, max_rotations=4, max_depth=64)
    rng = np.random.default_rng(21)
    gv = VecEnv("pauli", n, gateset, batch, **cfg)
    assert gv.pauli_num_perms() == 2
    envs = [OracleEnv("pauli", n, gateset, **{k: int(v) for k, v in cfg.items()}) for _ in range(batch)]
    tabs, labs = [], []
    for o in envs:  # each oracle env computes its own permutations (pauli.rs:374-378; oracle/qgym_oracle_symmetry.c)
        t = random_tableau(rng, n, 10, pairs)
        l = random_labels(rng, n, int(rng.integers(0, 5)), 3)
        o.pauli_reset_from(t, l)
        tabs.append(t)
        labs.append(l)
    gv.pauli_reset_from(np.stack(tabs), labs)
    for t in range(30):
        draws = rng.integers(0, 2, size=batch)
        got = gv.pauli_observe(torch.as_tensor(draws, device="cuda", dtype=torch.int32)).cpu().numpy()
        want = np.stack([o.dense_obs(int(d)) for o, d in zip(envs, draws)])
        np.testing.assert_array_equal(got, want, err_msg=f"obs t={t}")
        acts = rng.integers(0, A, size=batch)
        for o, a in zip(envs, acts):
            o.step(int(a))
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32))
        gv.sync()
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32))
        np.testing.assert_array_equal(gv.success.cpu().numpy(), [int(o.success()) for o in envs])
    for e in range(0, batch, 9):
        assert gv.solution(e) == envs[e].solution()  # the log holds the un-permuted (actual) actions


@pytest.mark.parametrize("n,difficulty,scale,max_rot", [(6, 40, 4, 5), (20, 96, 16, 5), (4, 3, 8, 5), (10, 400, 1, 26)])
def test_pauli_reset_generates_the_replayed_target(n, difficulty, scale, max_rot):
    """qg_vec_reset(seed) on a PauliEnv batch = PauliEnv::reset with its target generator
    (reference pauli.rs:54-271, 554-586) on the documented counter-RNG stream."""
    from qiskit_gym_amd.vec import VecEnv

    gateset = line_gateset("pauli", n)
    batch = 48
    cfg = dict(add_perms=False, track_solution=False, max_rotations=max_rot, difficulty=difficulty, pauli_diff_scale=scale)
    gv = VecEnv("pauli", n, gateset, batch, **cfg)
    gv.reset(seed=0xC0FFEE)
    gv.sync()
    envs = []
    for e in range(batch):
        o = OracleEnv("pauli", n, gateset, **{k: int(v) for k, v in cfg.items()})
        o.pauli_reset_seeded(0xC0FFEE, e)
        envs.append(o)
    np.testing.assert_array_equal(gv.observe().cpu().numpy(), np.stack([o.dense_obs() for o in envs]))
    np.testing.assert_array_equal(gv.depth.cpu().numpy(), [o.depth() for o in envs])
    np.testing.assert_array_equal(gv.success.cpu().numpy(), [int(o.success()) for o in envs])
    assert any(o.active_rotations() for o in envs) or difficulty // scale == 0
    if max_rot > 16:
        assert max(len(o.active_rotations()) for o in envs) > 16  # the generator filled more than the 16-rotation kernels hold
    # and the scalar Gym front-end can reset a PauliGym
    from qiskit_gym_amd.envs import PauliGym

    g = PauliGym.from_coupling_map([(i, i + 1) for i in range(n - 1)] + [(i + 1, i) for i in range(n - 1)], difficulty=difficulty, add_perms=False)
    obs, info = g.reset(seed=3)
    assert obs.shape == (2 * n, 2 * n + 5) and obs.dtype == np.int8  # PauliGym's own default max_rotations
