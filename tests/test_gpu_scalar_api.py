"""The scalar `Env`-trait flavour of the C ABI (qg_env_*) replaying the reference's own recorded
notebook transcripts (tests/golden/) on the GPU, plus clone / twists behaviour."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import OracleEnv  # noqa: E402
from qiskit_gym_amd.envs.raw import RawEnv  # noqa: E402
from util import line_gateset, grid_gateset  # noqa: E402


def load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def norm(gs):
    return [(n, tuple(q)) for n, q in gs]


def test_lf_notebook_transcripts_on_gpu(golden_dir):
    d = load(golden_dir, "lf_line3_transcripts.json")
    for seq in d["sequences"]:
        env = RawEnv("linear_function", d["num_qubits"], d["gateset"], add_inverts=False, add_perms=False)
        assert env.num_actions() == d["action_space_n"] and env.obs_shape() == d["obs_shape"]
        env.set_state(np.array(d["start_state"]).flatten().tolist())
        for a, want, fin in zip(seq["actions"], seq["states"], seq["is_final"]):
            assert not env.is_final()
            env.step(a)
            full = np.zeros(9, dtype=np.int8)
            full[env.observe()] = 1  # adapters.py:50-54
            assert full.reshape(3, 3).tolist() == want
            assert env.is_final() == fin


@pytest.mark.parametrize("key", ["permutation_swap_0_8", "linear_function_cx_0_4", "clifford_h_2"])
def test_recorded_synthesis_outputs_on_gpu(golden_dir, key):
    rec = load(golden_dir, "notebook_solutions.json")[key]
    gs = norm(rec["gateset"])
    env = RawEnv(rec["env"], rec["num_qubits"], gs, add_inverts=False, add_perms=False)
    env.set_state(np.array(rec["state"]).flatten().tolist())
    actions = [gs.index((n, tuple(q))) for n, q in rec["circuit"]]
    for a in actions:
        assert not env.is_final()
        env.step(a)
    assert env.success() and env.is_final() and env.reward() > 0.9
    assert env.solution() == actions
    assert env.masks() == [False] * len(gs)


def test_constructor_state_clone_and_inverts():
    gs = line_gateset("clifford", 4)
    env = RawEnv("clifford", 4, gs, add_perms=False)  # add_inverts defaults to True (clifford.rs:420)
    ora = OracleEnv("clifford", 4, gs, add_perms=0)
    # fresh env: identity, depth 1, success, reward 1.0 -> is_final (clifford.rs:214-245)
    assert env.is_final() and env.success() and env.reward() == 1.0
    rng = np.random.default_rng(0)
    state = rng.integers(0, 2, size=64)
    state = (np.eye(8, dtype=np.int64) + np.triu(state.reshape(8, 8), 1)) % 2  # unit upper triangular: invertible
    env.set_state(state.flatten().tolist())
    ora.set_state(state.flatten().tolist())
    clone = env.clone()
    for t in range(12):
        a, coin = int(rng.integers(len(gs))), int(rng.integers(2))
        env.step(a, coin)
        ora.step(a, coin)
        assert env.observe() == ora.observe()
        assert np.float32(env.reward()).view(np.uint32) == ora.reward_bits()
    assert env.solution() == ora.solution()
    # the clone still holds the state it was cloned at
    full = np.zeros(64, dtype=np.int64)
    full[clone.observe()] = 1
    assert full.tolist() == state.flatten().tolist()


@pytest.mark.parametrize("kind,n,inverts", [("clifford", 5, False), ("clifford", 20, False), ("linear_function", 12, False),
                                            ("linear_function", 40, False), ("clifford", 4, True), ("permutation", 9, False), ("pauli", 4, False),
                                            ("linear_function", 20, True), ("linear_function", 50, True), ("permutation", 30, True), ("clifford", 20, True)])
def test_clone_mid_episode_continues_like_the_original(kind, n, inverts):
    """Env: DynClone -- a clone taken mid-episode carries every piece of resident state (including the
    incremental solved mask of the one-step kernels) and then evolves independently."""
    gs = grid_gateset("permutation", 3, 3) if (kind == "permutation" and n == 9) else line_gateset(kind, n)
    kw = dict(add_perms=False, difficulty=6)
    if kind != "pauli":
        kw["add_inverts"] = inverts
    env = RawEnv(kind, n, gs, **kw)
    ora = OracleEnv(kind, n, gs, **{k: int(v) for k, v in kw.items()})
    rng = np.random.default_rng(3)
    if kind == "pauli":
        env.reset(5)
        ora.pauli_reset_seeded(5, 0)
    else:
        ora.reset_with(rng.integers(0, len(gs), size=6))
        start = ora.get_state().tolist()
        env.set_state(start)  # the same scrambled state through the trait's own entry point
        ora.set_state(start)
    for t in range(3):
        a, coin = int(rng.integers(len(gs))), int(rng.integers(2)) if inverts else 0
        env.step(a, coin) if kind != "pauli" else env.step(a)
        ora.step(a, coin)
    twin, ora_twin = env.clone(), ora.clone()
    for t in range(10):
        a, b = int(rng.integers(len(gs))), int(rng.integers(len(gs)))
        for e, o, act in ((env, ora, a), (twin, ora_twin, b)):
            coin = int(rng.integers(2)) if inverts else 0  # an inversion: transpose (Clifford), role swap (LinearFunction), scatter (Permutation)
            e.step(act, coin) if kind != "pauli" else e.step(act)
            o.step(act, coin)
            assert e.observe() == o.observe(), (kind, t)
            assert np.float32(e.reward()).view(np.uint32) == o.reward_bits(), (kind, t)
            assert e.success() == o.success() and e.is_final() == o.is_final()
    # a clone of a solved env reports success; a clone of an unsolved one must not
    assert twin.success() == ora_twin.success()


def test_twists_match_symmetry_rules():
    # line-3 CX+SWAP: automorphisms {id, reversal}; both map the gateset onto itself
    gs = line_gateset("linear_function", 3)
    env = RawEnv("linear_function", 3, gs, add_inverts=False)
    obs_perms, act_perms = env.twists()
    assert len(obs_perms) == 2 and len(act_perms) == 2
    # the gate index map keeps the LAST gate of each canonical key (symmetry.rs:217-223), so with
    # both SWAP(a,b) and SWAP(b,a) in the gateset even the identity twist maps SWAP(0,1) -> SWAP(1,0)
    def want_act(perm):
        out = []
        for name, (a, b) in gs:
            key = (name, tuple(sorted((perm[a], perm[b]))) if name == "SWAP" else (perm[a], perm[b]))
            out.append([i for i, (n2, q2) in enumerate(gs) if (n2, tuple(sorted(q2)) if n2 == "SWAP" else tuple(q2)) == key][-1])
        return out

    for k, perm in enumerate(([0, 1, 2], [2, 1, 0])):  # sorted automorphisms of the 3-line
        assert obs_perms[k] == [perm[r] * 3 + perm[c] for r in range(3) for c in range(3)]  # symmetry.rs:265-274
        assert act_perms[k] == want_act(perm)
    assert act_perms[0] == [0, 1, 2, 3, 5, 5, 7, 7]
    # custom 3q Clifford gateset with H/S only on qubit 0: only the identity survives
    cg = [("CX", (0, 1)), ("CX", (1, 0)), ("CX", (1, 2)), ("CX", (2, 1)), ("SWAP", (0, 1)), ("SWAP", (1, 2)), ("H", (0,)), ("S", (0,))]
    env = RawEnv("clifford", 3, cg, add_inverts=False)
    obs_perms, act_perms = env.twists()
    assert len(obs_perms) == 1 and obs_perms[0] == list(range(36))
    env = RawEnv("clifford", 3, cg, add_inverts=False, add_perms=False)
    assert env.twists() == ([], [])


@pytest.mark.parametrize("kind,n,edges", [("linear_function", 4, "line"), ("clifford", 4, "line"), ("clifford", 4, "ring"), ("permutation", 5, "line")])
def test_twists_are_symmetries_of_the_dynamics(kind, n, edges):
    """Env::twists (symmetry.rs:115-361): every (obs_perm, act_perm) pair must commute with stepping -- relabel the
    observation with obs_perm, the action with act_perm, and the relabelled env does the relabelled thing, with the same
    reward.  twisterl, which applies the twists, is absent from the reference tree; the convention under which the pairs
    are symmetries is obs'[obs_perm[i]] = obs[i] with action act_perm[a] (for the involutions of a line the direction does not
    matter; the rotations of a ring tell the two apart).  One convention must work for every twist, state and action."""
    from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map, line_edges
    from util import ALLOWED

    e = line_edges(n, True)
    if edges == "ring":
        e = e + [(n - 1, 0), (0, n - 1)]
    gs = gateset_from_coupling_map(e, None, ALLOWED[kind])[1]
    env = RawEnv(kind, n, gs, add_inverts=False, add_perms=True, track_solution=False, difficulty=6)
    obs_perms, act_perms = env.twists()
    assert len(obs_perms) == len(act_perms) >= 2 and sorted(obs_perms[0]) == list(range(len(obs_perms[0])))
    size = len(obs_perms[0])

    def dense(e_):
        d = np.zeros(size, dtype=np.int64)
        d[np.asarray(e_.observe(), dtype=np.int64)] = 1
        return d

    def state_of(d):  # set_state wire format: the dense 0/1 matrix (Clifford / LinearFunction) or the permutation itself
        return d.reshape(n, n).argmax(axis=1).tolist() if kind == "permutation" else d.tolist()

    rng = np.random.default_rng(n)
    twin = RawEnv(kind, n, gs, add_inverts=False, add_perms=True, track_solution=False, difficulty=6)
    for trial in range(12):
        env.reset(100 + trial)
        for k, (op, ap) in enumerate(zip(obs_perms, act_perms)):
            d = dense(env)
            op = np.asarray(op)

            def relabel(x):
                y = np.zeros_like(x)
                y[op] = x
                return y

            twin.set_state(state_of(relabel(d)))
            assert dense(twin).tolist() == relabel(d).tolist()
            a = int(rng.integers(0, len(gs)))
            probe = env.clone()
            probe.step(a)
            twin.step(ap[a])
            assert dense(twin).tolist() == relabel(dense(probe)).tolist(), (kind, k, a)
            # same gate cost (the solved bonus may differ: set_state gives the twin max_depth, not the probe's depth)
            assert twin.success() == probe.success()


def test_destroyed_envs_are_pooled_for_clones_and_the_pool_can_be_released():
    """qg_env_destroy parks a handle for the next clone of the same configuration (<= 64 per configuration); a clone taken from the pool
    starts from its source's state whatever its previous owner did, and qg_env_pool_clear hands everything back."""
    from qiskit_gym_amd.envs.raw import RawEnv

    gs = line_gateset("clifford", 4)
    proto = RawEnv("clifford", 4, gs, add_inverts=False, add_perms=False, track_solution=True, difficulty=6, seed=3)
    proto.reset()
    scratch = [proto.clone() for _ in range(70)]
    for i, c in enumerate(scratch):  # their previous lives differ
        for a in range(i % 5):
            c.step(a)
    del scratch, c  # 70 destroyed handles: 64 parked, the rest freed
    want = (proto.observe(), proto.reward(), proto.is_final(), proto.solution())
    clones = [proto.clone() for _ in range(66)]  # 64 from the pool, two fresh
    for c in clones:
        assert (c.observe(), c.reward(), c.is_final(), c.solution()) == want
        c.step(1)
    proto.step(1)
    assert clones[0].observe() == proto.observe() and clones[-1].observe() == proto.observe()
    del clones
    RawEnv.release_pool()
    again = proto.clone()  # the pool is empty: a fresh handle, same answer
    assert again.observe() == proto.observe()
