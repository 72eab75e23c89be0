"""Edge cases: maximum supported sizes, tiny sizes, empty gatesets, ragged batches, limits that
must fail loudly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import OracleEnv, OracleVec  # noqa: E402
from qiskit_gym_amd._lib import QGymError  # noqa: E402
from test_gpu_pauli import random_labels, random_tableau  # noqa: E402
from util import f32_bits, grid_gateset, line_gateset, make_pair  # noqa: E402


def _run(kind, n, gs, B, steps, inverts, per_env):
    A = len(gs)
    ov, gv = make_pair(kind, n, gs, B, add_inverts=inverts, add_perms=False, track_solution=True, difficulty=2 * n, max_depth=96)
    rng = np.random.default_rng(n + B)
    draws = rng.integers(0, A, size=(2 * n, B))
    ov.reset_with(draws)
    gv.reset_with(torch.as_tensor(draws, device="cuda", dtype=torch.int32))
    for t in range(steps):
        acts = rng.integers(0, A, size=B)
        coins = rng.integers(0, 2, size=B) if inverts else None
        r, s, f, d = ov.step(acts, coins)
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int64),
                None if coins is None else torch.as_tensor(coins, device="cuda", dtype=torch.uint8))
        gv.sync()
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r), err_msg=f"t={t}")
        np.testing.assert_array_equal(gv.done.cpu().numpy(), f)
    np.testing.assert_array_equal(gv.get_state("i64").cpu().numpy(), ov.get_state(per_env))
    np.testing.assert_array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense())
    assert gv.solution(B - 1) == ov.env(B - 1).solution()


@pytest.mark.parametrize("inverts", [False, True])
def test_maximum_sizes(inverts):
    _run("clifford", 32, line_gateset("clifford", 32), 70, 12, inverts, 64 * 64)          # 64 uint64 rows
    _run("linear_function", 64, line_gateset("linear_function", 64), 66, 12, inverts, 64 * 64)
    _run("linear_function", 32, line_gateset("linear_function", 32), 129, 12, inverts, 32 * 32)  # widest TILE / ROWS32 shape
    _run("permutation", 16, grid_gateset("permutation", 4, 4), 65, 20, inverts, 16)


@pytest.mark.parametrize("kind", ["clifford", "linear_function", "permutation"])
def test_single_qubit_and_single_env(kind):
    gs = [("SWAP", (0, 0))] if kind == "permutation" else ([("CX", (0, 0)), ("SWAP", (0, 0))] if kind == "linear_function" else
                                                          [("H", (0,)), ("S", (0,)), ("SX", (0,)), ("CX", (0, 0)), ("CZ", (0, 0)), ("SWAP", (0, 0))])
    _run(kind, 1, gs, 1, 6, False, {"clifford": 4, "linear_function": 1, "permutation": 1}[kind])


def test_empty_gateset_steps_are_noops_and_reset_panics():
    from qiskit_gym_amd.vec import VecEnv

    gv = VecEnv("clifford", 3, [], 5, add_inverts=False, add_perms=False, track_solution=False)
    o = OracleEnv("clifford", 3, [], add_inverts=0, add_perms=0, track_solution=0)
    assert gv.num_actions() == 0
    st = np.eye(6, dtype=np.int64)
    st[0, 3] = 1
    gv.set_state(np.tile(st.reshape(1, -1), (5, 1)), "i64")
    o.set_state(st.reshape(-1).tolist())
    for _ in range(3):  # every action is out of range: no gate, depth still decrements (clifford.rs:324,342)
        gv.step(torch.zeros(5, dtype=torch.int32, device="cuda"))
        o.step(0)
    gv.sync()
    assert int(gv.depth[0]) == o.depth() == 125 and float(gv.reward[0]) == o.reward() == 0.0
    with pytest.raises(QGymError):  # Uniform::new(0, 0) panics in the reference's reset()
        gv.reset(1)


@pytest.mark.parametrize("rot", [16, 32])
def test_pauli_maximum_size_32_qubits(rot):
    from qiskit_gym_amd.vec import VecEnv

    n, B = 32, 40
    gs = line_gateset("pauli", n)
    A = len(gs)
    pairs = [g[1] for g in gs if g[0] == "CX"]
    cfg = dict(add_perms=False, track_solution=True, max_rotations=rot, final_pauli_layers=rot, max_depth=64)
    rng = np.random.default_rng(9)
    gv = VecEnv("pauli", n, gs, B, **cfg)
    envs = [OracleEnv("pauli", n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(B)]
    tabs, labs = [], []
    for o in envs:
        t = random_tableau(rng, n, 40, pairs)
        l = random_labels(rng, n, rot, 3)
        o.pauli_reset_from(t, l)
        tabs.append(t)
        labs.append(l)
    gv.pauli_reset_from(np.stack(tabs), labs)
    for t in range(40):
        acts = rng.integers(0, A, size=B)
        for o, a in zip(envs, acts):
            o.step(int(a))
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32))
        gv.sync()
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32), err_msg=f"t={t}")
    np.testing.assert_array_equal(gv.observe().cpu().numpy(), np.stack([o.dense_obs() for o in envs]))
    for e in range(0, B, 5):
        assert gv.solution(e) == envs[e].solution()


def test_limits_fail_loudly():
    from qiskit_gym_amd.vec import VecEnv

    with pytest.raises(QGymError, match="N <= 32"):
        VecEnv("clifford", 33, line_gateset("clifford", 33), 4)
    with pytest.raises(QGymError, match="N <= 64"):
        VecEnv("linear_function", 65, line_gateset("linear_function", 65), 4)
    with pytest.raises(QGymError, match="N <= 256"):
        VecEnv("permutation", 257, [("SWAP", (0, 256))], 4)
    with pytest.raises(QGymError, match="rotations"):
        VecEnv("pauli", 4, line_gateset("pauli", 4), 4, max_rotations=33)
    with pytest.raises(QGymError, match="out of range"):
        VecEnv("clifford", 3, [("H", (3,))], 4)
    with pytest.raises(QGymError, match="batch"):
        VecEnv("clifford", 3, [("H", (0,))], 0)


def test_pauli_weight_zero_rotation_is_reported():
    """An all-identity rotation reaches the front layer with weight 0: the reference's
    `which_qubit(..).unwrap()` panics (pauli_network.rs:113-114); here the env gets a fault bit."""
    from qiskit_gym_amd.vec import VecEnv

    n = 3
    gs = line_gateset("pauli", n)
    gv = VecEnv("pauli", n, gs, 2, add_perms=False, track_solution=False)
    state = [1] + np.eye(2 * n, dtype=np.int64).reshape(-1).tolist() + [n] + [ord("I")] * n
    gv.set_state(np.array([state, state], dtype=np.int64), "i64")
    cx = [i for i, g in enumerate(gs) if g[0] == "CX"][0]
    gv.step(torch.full((2,), cx, dtype=torch.int32, device="cuda"))
    with pytest.raises(QGymError, match="weight-0"):
        gv.sync()
    o = OracleEnv("pauli", n, gs, add_perms=0, track_solution=0)
    o.set_state(state)
    from oracle import OracleError

    with pytest.raises(OracleError):
        o.step(cx)


def test_four_million_envs_index_arithmetic():
    """B = 2^22 CliffordGym 16q envs (512 MiB of state, a 4 GiB dense observation: element counts beyond
    2^32) -- every kernel's index arithmetic at a size no other test reaches; a strided sample of envs is
    checked against the oracle."""
    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv
    from util import rng_actions

    n, B, T, diff = 16, 1 << 22, 6, 24
    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    gv = VecEnv("clifford", n, gs, B, **cfg)
    gv.reset(77)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda", generator=gen)
    for t in range(T):
        gv.step(acts[t])
    gv.sync()
    ids = np.concatenate([np.arange(0, 64), np.arange(B - 64, B), np.arange(12345, B, 65537)])
    proto = OracleEnv("clifford", n, gs, **{k: int(v) for k, v in cfg.items()})
    ov = OracleVec(proto, len(ids))
    ov.reset_with(rng_actions(77, ids, diff, A))
    acts_h = acts[:, torch.as_tensor(ids, device="cuda")].cpu().numpy()
    for t in range(T):
        r, s, f, d = ov.step(acts_h[t])
    idx = torch.as_tensor(ids, device="cuda")
    np.testing.assert_array_equal(f32_bits(gv.reward[idx].cpu().numpy()), f32_bits(r))
    np.testing.assert_array_equal(gv.depth[idx].cpu().numpy(), d)
    obs = gv.observe()  # [B, 32, 32] int8 = 4 GiB
    np.testing.assert_array_equal(obs[idx].cpu().numpy().reshape(len(ids), -1), ov.observe_dense())
    del obs
    x = gv.observe_as(torch.bfloat16)  # 8 GiB
    assert torch.equal(x[idx].float(), torch.as_tensor(ov.observe_dense(), device="cuda").float())
    del x
    packed = gv.observe_packed()
    want = (ov.observe_dense().reshape(len(ids), 32, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)
    np.testing.assert_array_equal(packed[idx].cpu().numpy().view(np.uint32), want)
    # reset_done over the whole batch (list compaction with 2^22 entries)
    gv.done.fill_(1)
    gv.reset_done(9)
    gv.sync()
    ov.reset_with(rng_actions(9, ids, diff, A))
    np.testing.assert_array_equal(gv.observe_packed()[idx].cpu().numpy().view(np.uint32),
                                  (ov.observe_dense().reshape(len(ids), 32, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32))
    gv.close()


def test_four_million_pauli_envs_index_arithmetic():
    """B = 2^22 PauliGym 16q envs: the ragged observation expansion beyond 2^32 elements, the device-side target
    generator and the compacted reset at that size; a strided sample is checked against the oracle."""
    from qiskit_gym_amd.vec import VecEnv

    n, B, T = 16, 1 << 22, 5
    gs = line_gateset("pauli", n)
    A = len(gs)
    cfg = dict(add_perms=False, track_solution=False, max_rotations=5, difficulty=24, pauli_diff_scale=4)
    gv = VecEnv("pauli", n, gs, B, **cfg)
    rows, cols = gv.obs_shape_
    assert B * rows * cols > 2**32
    gv.reset(31)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(6)
    acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda", generator=gen)
    for t in range(T):
        gv.step(acts[t])
    gv.sync()
    ids = np.concatenate([np.arange(0, 40), np.arange(B - 40, B), np.arange(999, B, 262147)])
    envs = [OracleEnv("pauli", n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in ids]
    acts_h = acts[:, torch.as_tensor(ids, device="cuda")].cpu().numpy()
    for o, e, col in zip(envs, ids, acts_h.T):
        o.pauli_reset_seeded(31, int(e))
        for a in col:
            o.step(int(a))
    idx = torch.as_tensor(ids, device="cuda")
    np.testing.assert_array_equal(f32_bits(gv.reward[idx].cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32))
    obs = gv.observe()
    np.testing.assert_array_equal(obs[idx].cpu().numpy(), np.stack([o.dense_obs() for o in envs]))
    del obs
    gv.done.fill_(1)
    gv.reset_done(8)
    gv.sync()
    for o, e in zip(envs, ids):
        o.pauli_reset_seeded(8, int(e))
    np.testing.assert_array_equal(gv.observe_as(torch.float16)[idx].float().cpu().numpy().reshape(len(ids), rows, cols),
                                  np.stack([o.dense_obs() for o in envs]).astype(np.float32))
    gv.close()


@pytest.mark.parametrize("n", [9, 25])  # nibble-packed and byte-per-entry layouts
def test_permutation_set_state_with_repeated_entries_runs_like_the_reference(n):
    """permutation.rs:168-173 stores the vector unvalidated; step / observe / solved work on any in-range vector.  Without add_inverts
    such a state steps like the oracle's; with add_inverts it has no inverse, and both layouts report the fault at set_state."""
    side = int(round(n ** 0.5))
    gs = grid_gateset("permutation", side, side)
    A, B = len(gs), 200
    rng = np.random.default_rng(n)
    states = np.stack([rng.permutation(n) for _ in range(B)]).astype(np.int64)
    states[::2, 1] = states[::2, 0]  # every other env holds one entry twice
    ov, gv = make_pair("permutation", n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=4)
    ov.set_state(states)
    gv.set_state(states)
    gv.sync()  # no fault
    for t in range(12):
        acts = rng.integers(0, A, size=B)
        r, s, f, d = ov.step(acts)
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32))
        gv.sync()
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r))
        np.testing.assert_array_equal(gv.success.cpu().numpy(), s)
        np.testing.assert_array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense())
    from qiskit_gym_amd.vec import VecEnv

    inv = VecEnv("permutation", n, gs, B, add_inverts=True, add_perms=False, track_solution=False, difficulty=4)
    inv.set_state(states)
    with pytest.raises(QGymError):
        inv.sync()


@pytest.mark.parametrize("B,inverts,act64", [(1003, False, False), (1, True, True), (4096, True, False), (77, False, True)])
def test_step_host_from_pinned_and_pageable_memory_equals_the_device_step(B, inverts, act64):
    """qg_vec_step_host: pinned, device-mapped buffers take the zero-copy path (the step reads the actions in host memory, one kernel
    writes reward / is_final / success there; ragged tails, a batch of one), pageable buffers the copies; both must equal qg_vec_step."""
    from qiskit_gym_amd import _lib
    from qiskit_gym_amd.vec import VecEnv

    n, T = 7, 5
    gs = line_gateset("clifford", n)
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=False, difficulty=5)
    ref, pin, page = (VecEnv("clifford", n, gs, B, seed=2, **cfg) for _ in range(3))
    L = ref._L
    rng = np.random.default_rng(B)
    adt = np.int64 if act64 else np.int32
    code = 1 if act64 else 0  # QG_ACT_I64 / QG_ACT_I32
    for env in (ref, pin, page):
        env.reset(11)
    pinned = {k: torch.empty(B, dtype=dt).pin_memory() for k, dt in (("a", torch.int64 if act64 else torch.int32), ("c", torch.uint8), ("r", torch.float32),
                                                                     ("d", torch.uint8), ("s", torch.uint8))}
    for t in range(T):
        acts = rng.integers(-1, len(gs) + 1, size=B).astype(adt)
        coins = rng.integers(0, 2, size=B).astype(np.uint8)
        ref.step(torch.as_tensor(acts, device="cuda"), torch.as_tensor(coins, device="cuda") if inverts else None)
        ref.sync()
        want = (ref.reward.cpu().numpy().view(np.uint32), ref.done.cpu().numpy(), ref.success.cpu().numpy())
        # pinned
        pinned["a"].numpy()[:] = acts
        pinned["c"].numpy()[:] = coins
        for k in "rds":
            pinned[k].numpy()[:] = 0x55 if k != "r" else -1.0
        _lib.check(L.qg_vec_step_host(pin._h, pinned["a"].data_ptr(), code, pinned["c"].data_ptr() if inverts else None, pinned["r"].data_ptr(),
                                      pinned["d"].data_ptr(), pinned["s"].data_ptr(), pin._stream()))
        pin.sync()
        got = (pinned["r"].numpy().view(np.uint32), pinned["d"].numpy(), pinned["s"].numpy())
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w, err_msg=f"pinned, t={t}")
        # pageable
        r, d, s = np.full(B, -1.0, np.float32), np.full(B, 0x55, np.uint8), np.full(B, 0x55, np.uint8)
        _lib.check(L.qg_vec_step_host(page._h, acts.ctypes.data, code, coins.ctypes.data if inverts else None, r.ctypes.data, d.ctypes.data, s.ctypes.data,
                                      page._stream()))
        page.sync()
        for g, w in zip((r.view(np.uint32), d, s), want):
            np.testing.assert_array_equal(g, w, err_msg=f"pageable, t={t}")
    assert torch.equal(pin.get_state("packed"), ref.get_state("packed")) and torch.equal(page.get_state("packed"), ref.get_state("packed"))


@pytest.mark.parametrize("kind,n,B", [("clifford", 16, 1003), ("clifford", 3, 77), ("linear_function", 5, 4096), ("clifford", 9, 1), ("permutation", 12, 333)])
def test_masks_are_all_ones_except_for_solved_envs(kind, n, B):
    """Env::masks (clifford.rs:349-351): every action allowed unless the env is solved -- exact for every (env, action) byte; action counts
    above and below 16 (the 16-bytes-per-thread kernel and the byte kernel), ragged totals, a batch of one."""
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset(kind, n)
    cfg = dict(add_perms=False, track_solution=False, difficulty=1)
    if kind != "pauli":
        cfg["add_inverts"] = False
    env = VecEnv(kind, n, gs, B, **cfg)
    env.reset(5)  # difficulty 1: one random gate per env, a good share of them is solved again after one more
    g = torch.Generator(device="cuda").manual_seed(B)
    env.step(torch.randint(0, len(gs), (B,), dtype=torch.int32, device="cuda", generator=g))
    env.sync()
    suc = env.success.cpu().numpy()
    m = env.masks().cpu().numpy()
    assert m.shape == (B, len(gs))
    np.testing.assert_array_equal(m, np.repeat((suc == 0).astype(np.uint8)[:, None], len(gs), axis=1))
    if B > 50:
        assert 0 < suc.sum() < B


@pytest.mark.parametrize("kind,n", [("clifford", 17), ("clifford", 24), ("clifford", 32), ("linear_function", 20), ("linear_function", 33), ("linear_function", 64)])
def test_dense_observation_of_the_wide_layouts_equals_the_packed_rows(kind, n):
    """observe() of the 64-bit-row and lane-group layouts goes packed rows -> expansion kernel: it must be the bits of observe_packed(), and
    the oracle's observation on a sample."""
    from qiskit_gym_amd.vec import VecEnv

    B = 517
    gs = line_gateset(kind, n)
    cfg = dict(add_inverts=(kind == "linear_function"), add_perms=False, track_solution=False, difficulty=3 * n)
    ov, gv = make_pair(kind, n, gs, B, **cfg)
    rng = np.random.default_rng(n)
    draws = rng.integers(0, len(gs), size=(3 * n, B))
    ov.reset_with(draws)
    gv.reset_with(torch.as_tensor(draws, device="cuda", dtype=torch.int32))
    for t in range(3):
        acts = rng.integers(0, len(gs), size=B)
        coins = rng.integers(0, 2, size=B) if cfg["add_inverts"] else None
        ov.step(acts, coins)
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32), None if coins is None else torch.as_tensor(coins, device="cuda", dtype=torch.uint8))
    gv.sync()
    dense = gv.observe().cpu().numpy().reshape(B, -1)
    np.testing.assert_array_equal(dense, ov.observe_dense())
    rows = gv.observe_packed().cpu().numpy()
    D = dense.shape[1] // rows.shape[1]
    bits = ((rows.astype(np.uint64)[:, :, None] >> np.arange(D, dtype=np.uint64)) & 1).astype(np.int8).reshape(B, -1)
    np.testing.assert_array_equal(dense, bits)
