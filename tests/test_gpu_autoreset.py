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


@pytest.mark.parametrize("n,max_rot,coupling", [(20, 8, "line"), (12, 4, "line"), (28, 16, "line"), (13, 8, "all"), (16, 8, "grid")])
def test_pauli_reset_done_of_a_short_list_runs_as_a_tree_and_equals_the_per_lane_generator_and_the_oracle(n, max_rot, coupling):
    """A few finished envs of a large batch with a long tableau scramble: a workgroup per env (ptile_reset_tree_kernel: every thread draws one
    gate, the chain is cut in four and multiplied back).  A fresh target depends on (seed, env) only, so the same envs regenerated with many
    others finished too (the per-lane generator) must come out identical, and so must the oracle's."""
    from qiskit_gym_amd.vec import VecEnv

    B, few, many = 8192, 37, 900  # (lists are compacted above 4 096 envs)
    if coupling == "all":  # one distance class of n (n - 1) / 2 = 78 pairs: more than a wave of lanes (the tree's label generator walks it in chunks)
        from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map
        from util import ALLOWED
        gs = gateset_from_coupling_map([(i, j) for i in range(n) for j in range(n) if i != j], None, ALLOWED["pauli"])[1]
    elif coupling == "grid":
        gs = grid_gateset("pauli", 4, 4, bidirectional=True)
    else:
        gs = line_gateset("pauli", n)
    cfg = dict(add_perms=False, track_solution=False, max_rotations=max_rot, difficulty=128, pauli_diff_scale=16, depth_slope=1, max_depth=200)
    a, b = VecEnv("pauli", n, gs, B, **cfg), VecEnv("pauli", n, gs, B, **cfg)
    a.reset(3)
    b.reset(3)
    rng = np.random.default_rng(n)
    listed = np.sort(rng.choice(B, size=few, replace=False))
    others = np.setdiff1d(np.arange(B), listed)
    extra = rng.choice(others, size=many, replace=False)
    before = a.observe().clone()
    a.done.zero_()
    b.done.zero_()
    a.done[torch.as_tensor(listed, device="cuda")] = 1
    b.done[torch.as_tensor(np.concatenate([listed, extra]), device="cuda")] = 1
    a.reset_done(77)   # 37 <= B / 32: the tree
    b.reset_done(77)   # 937 > B / 32: one lane per env
    a.sync()
    b.sync()
    oa, ob = a.observe(), b.observe()
    li = torch.as_tensor(listed, device="cuda")
    assert torch.equal(oa[li], ob[li])
    for name in ("depth", "done", "success"):
        assert torch.equal(getattr(a, name)[li], getattr(b, name)[li]), name
    assert torch.equal(a.reward[li].view(torch.int32), b.reward[li].view(torch.int32))
    keep = torch.as_tensor(others, device="cuda")
    assert torch.equal(oa[keep], before[keep])  # live episodes are untouched
    assert not torch.equal(oa[li], before[li])
    for e in listed[:6]:
        o = OracleEnv("pauli", n, gs, **{k: int(v) for k, v in cfg.items()})
        o.pauli_reset_seeded(77, int(e))
        np.testing.assert_array_equal(oa[int(e)].cpu().numpy(), o.dense_obs())
        assert int(a.depth[int(e)]) == o.depth() and bool(a.done[int(e)]) == o.is_final()


@pytest.mark.parametrize("kind,n", [("clifford", 16), ("clifford", 20), ("linear_function", 12)])
def test_reset_done_of_a_list_longer_than_the_tree_grid_is_walked_in_rounds(kind, n):
    """Trees take lists up to 4 096 envs with min(1 024, B / 8) workgroups: a longer list is walked in rounds (entry i on workgroup i mod grid).
    The same envs reset with even more finished beside them (beyond the trees' range: one lane per env) must come out identical -- a fresh
    episode depends on (seed, env) only -- and a few are replayed on the oracle."""
    from qiskit_gym_amd.vec import VecEnv

    B, listed_n, extra_n = 8192, 3000, 2500  # 3 000 -> three rounds of 1 024 workgroups; 5 500 -> not a tree
    gs = line_gateset(kind, n)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=80)
    a, b = VecEnv(kind, n, gs, B, **cfg), VecEnv(kind, n, gs, B, **cfg)
    a.reset(3)
    b.reset(3)
    rng = np.random.default_rng(n)
    perm = rng.permutation(B)
    listed, extra = np.sort(perm[:listed_n]), perm[listed_n:listed_n + extra_n]
    before = a.get_state("packed").clone()
    a.done.zero_()
    b.done.zero_()
    a.done[torch.as_tensor(listed, device="cuda")] = 1
    b.done[torch.as_tensor(np.concatenate([listed, extra]), device="cuda")] = 1
    a.reset_done(41)
    b.reset_done(41)
    a.sync()
    b.sync()
    sa, sb = a.get_state("packed"), b.get_state("packed")
    li = torch.as_tensor(listed, device="cuda")
    assert torch.equal(sa[li], sb[li])
    for name in ("depth", "done", "success"):
        assert torch.equal(getattr(a, name)[li], getattr(b, name)[li]), name
    keep = torch.as_tensor(np.setdiff1d(np.arange(B), listed), device="cuda")
    assert torch.equal(sa[keep], before[keep])  # live episodes are untouched
    dense = a.observe()
    draws = rng_actions(41, B, cfg["difficulty"], len(gs))
    for e in (listed[0], listed[1024], listed[2048], listed[-1]):  # envs of every round
        o = OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()})
        o.reset_with(draws[:, int(e)])
        np.testing.assert_array_equal(dense[int(e)].cpu().numpy(), o.dense_obs())


def _twin(kind, n, B, **cfg):
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset(kind, n)
    return gs, VecEnv(kind, n, gs, B, **cfg), VecEnv(kind, n, gs, B, **cfg)


def _same(a, b):
    a.sync()
    b.sync()
    assert torch.equal(a.get_state("packed"), b.get_state("packed"))
    assert torch.equal(a.depth, b.depth) and torch.equal(a.done, b.done) and torch.equal(a.reward.view(torch.int32), b.reward.view(torch.int32))


@pytest.mark.parametrize("kind,n,inverts", [("clifford", 16, False), ("clifford", 16, True), ("clifford", 20, False)])
def test_done_list_survives_graph_replays_of_single_steps(kind, n, inverts):
    """The list of finished envs is a device-side fact: a single step captured into a caller's graph appends to it on EVERY replay, and the
    host only sees the capture.  A graph of lone steps replayed many times, eager reset_done calls between replays, and a graph that opens
    with reset_done must all behave like the same calls made eagerly on a twin handle (which never saw a capture)."""
    B, diff = 2048, 2  # episodes of 4 steps: every replay finishes envs
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=False, difficulty=diff)
    gs, g, e = _twin(kind, n, B, **cfg)
    A = len(gs)
    gen = torch.Generator(device="cuda").manual_seed(12)
    acts = torch.randint(0, A, (3, B), dtype=torch.int32, device="cuda", generator=gen)
    coins = torch.randint(0, 2, (3, B), dtype=torch.uint8, device="cuda", generator=gen) if inverts else [None] * 3
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for h in (g, e):
            h.reset(3)
            h.step(acts[0], coins[0])
            h.reset_done(50)  # from here on single steps leave the list of the envs they finish
        # (1) lone steps in a graph, replayed 40 times: 40 x 3 steps append up to 120 x B entries unless every replay starts the list again
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for t in range(3):
                g.step(acts[t], coins[t])
        for _ in range(40):
            graph.replay()
        for _ in range(40):
            for t in range(3):
                e.step(acts[t], coins[t])
        _same(g, e)
        # (2) eager reset_done after replays, then replay again, then eager reset_done: no stale or duplicated entries
        for k in range(3):
            g.reset_done(100 + k)
            e.reset_done(100 + k)
            _same(g, e)
            graph.replay()
            for t in range(3):
                e.step(acts[t], coins[t])
            _same(g, e)
        # (3) a graph that opens with reset_done (the flags were set by launches outside it) and alternates step / reset_done
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=stream):
            for t in range(3):
                g.reset_done(200 + t)
                g.step(acts[t], coins[t])
        for rep in range(5):
            g2.replay()
            for t in range(3):
                e.reset_done(200 + t)
                e.step(acts[t], coins[t])
            _same(g, e)
            if rep == 2:  # an eager whole-batch reset between replays
                g.reset(9)
                e.reset(9)
        # (4) eager calls after captures still work (they trust nothing any more)
        for t in range(3):
            g.step(acts[t], coins[t])
            g.reset_done(300 + t)
            e.step(acts[t], coins[t])
            e.reset_done(300 + t)
        _same(g, e)


@pytest.mark.parametrize("kind,n,diff", [("clifford", 16, 256), ("clifford", 6, 101), ("clifford", 11, 64), ("linear_function", 20, 130), ("linear_function", 32, 67)])
def test_reset_done_of_a_few_envs_with_long_scrambles_matches_the_oracle(kind, n, diff):
    """Short lists of long scrambles take scramble_tree (a workgroup per env, the matrix by columns, the gate sequence cut in four and
    multiplied back as GF(2) matrices): the state it leaves must be the oracle's identity + `difficulty` gates drawn by global env id, for
    list lengths on both sides of its limits, with the other envs untouched."""
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset(kind, n)
    A, B, base = len(gs), 8192, 1000
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    gv = VecEnv(kind, n, gs, B, env_base=base, **cfg)
    gv.reset(7)
    ref = gv.get_state("packed").clone()
    rng = np.random.default_rng(diff)
    for count in (1, 37, 256, 300):  # 256 = B / 32 is the longest list the short-list kernels take; 300 goes to the thread-per-env path
        ids = np.sort(rng.choice(B, size=count, replace=False))
        gv.done[:] = 0
        gv.done[torch.as_tensor(ids, device="cuda")] = 1
        seed = 9000 + count
        gv.reset_done(seed)
        gv.sync()
        proto = OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()})
        from oracle import OracleVec
        ov = OracleVec(proto, count)
        ov.reset_with(rng_actions(seed, base + ids, diff, A))
        D = 2 * n if kind == "clifford" else n
        got = gv.get_state("i64").cpu().numpy()
        np.testing.assert_array_equal(got[ids], ov.get_state(D * D), err_msg=f"count={count}")
        mask = np.ones(B, dtype=bool)
        mask[ids] = False
        packed = gv.get_state("packed")
        assert torch.equal(packed[torch.as_tensor(mask, device="cuda")], ref[torch.as_tensor(mask, device="cuda")]), "a live env was touched"
        ref = packed.clone()
        assert (gv.depth.cpu().numpy()[ids] == min(2 * diff, 128)).all()  # clifford.rs:317


@pytest.mark.parametrize("case", range(12))
def test_reset_done_random_shapes_against_the_oracle(case):
    """Random env kind / size / difficulty / batch / fraction of finished envs: whichever reset path the library picks (thread per env,
    16 lanes per env, a workgroup per env with the product tree; 32- and 64-bit rows) must leave the oracle's state for the finished envs
    and nothing else.  QGYM_FUZZ_SEED_OFFSET shifts the cases."""
    import os

    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv

    rng = np.random.default_rng(4242 + case + int(os.environ.get("QGYM_FUZZ_SEED_OFFSET", "0")))
    kind = ["clifford", "linear_function"][int(rng.integers(0, 2))]
    n = int(rng.integers(3, 33)) if kind == "clifford" else int(rng.integers(9, 65))
    diff = int(rng.choice([1, 5, 40, 64, 65, 100, 200, 256]))
    B = int(rng.choice([64, 1000, 4096, 20000]))
    gs = line_gateset(kind, n)
    A = len(gs)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    base = int(rng.integers(0, 1 << 20))
    gv = VecEnv(kind, n, gs, B, env_base=base, **cfg)
    gv.reset(3)
    D = 2 * n if kind == "clifford" else n
    for rep in range(3):
        frac = float(rng.choice([0.002, 0.02, 0.1, 0.6]))
        count = max(1, min(B, int(B * frac)))
        ids = np.sort(rng.choice(B, size=count, replace=False))
        before = gv.get_state("packed").clone()
        gv.done[:] = 0
        gv.done[torch.as_tensor(ids, device="cuda")] = 1
        seed = int(rng.integers(0, 1 << 40))
        gv.reset_done(seed)
        gv.sync()
        ov = OracleVec(OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}), count)
        ov.reset_with(rng_actions(seed, base + ids, diff, A))
        got = gv.get_state("i64").cpu().numpy()
        np.testing.assert_array_equal(got[ids], ov.get_state(D * D), err_msg=f"{kind} {n}q diff {diff} B {B} count {count}")
        mask = torch.ones(B, dtype=torch.bool, device="cuda")
        mask[torch.as_tensor(ids, device="cuda")] = False
        assert torch.equal(gv.get_state("packed")[mask], before[mask]), "a live env was touched"


@pytest.mark.parametrize("kind,n,B,diff,track,dense", [
    ("clifford", 16, 700, 3, False, False),      # short scrambles: the flat / cooperative list paths inside the fused launch
    ("clifford", 16, 4096, 70, True, False),     # long scrambles, short lists: scramble_tree; solution log
    ("clifford", 16, 4096, 70, False, True),     # ... with a resident dense observation kept current by both halves of the launch
    ("clifford", 5, 333, 4, True, False),
    ("linear_function", 12, 1000, 5, False, False),
    ("linear_function", 32, 2048, 66, False, True),
    ("clifford", 24, 4096, 70, False, False),    # 64-bit rows (q64_reset_step_kernel): trees
    ("clifford", 20, 777, 3, True, False),       # ... short scrambles: the 16-lane / per-lane reset workgroups behind the trees and the steps; solution log
    ("linear_function", 40, 3000, 66, False, False),
    ("clifford", 32, 8192, 70, True, False),
])
def test_reset_done_step_in_one_launch_equals_the_two_calls(kind, n, B, diff, track, dense):
    """qg_vec_reset_done_step (from its second call on: ONE launch whose grid holds the reset's and the step's workgroups) against
    qg_vec_reset_done + qg_vec_step on a twin handle and against the oracle: rewards, flags, depth after every step, states, solution logs
    and the tracked dense observation at the end; episodes end at different times (depth 2 * difficulty, or earlier on success)."""
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset(kind, n)
    A = len(gs)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=track, difficulty=diff, depth_slope=1 if diff > 20 else 2, max_depth=128)
    fused, twin = VecEnv(kind, n, gs, B, **cfg), VecEnv(kind, n, gs, B, **cfg)
    envs = [OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(min(B, 400))]
    dt = fused.track_dense() if dense else None
    seed0 = 5
    fused.reset(seed0)
    twin.reset(seed0)
    draws = rng_actions(seed0, len(envs), diff, A)
    for e, o in enumerate(envs):
        o.reset_with(draws[:, e])
    rng = np.random.default_rng(B + diff)
    resets = 0
    if diff > 20:  # long episodes end together: spread the ends first (class env % 64 == k starts a new episode at prologue step k)
        cls = torch.arange(B, device="cuda") % 64
        for k in range(64):
            acts = rng.integers(0, A, size=B)
            ta = torch.as_tensor(acts, device="cuda", dtype=torch.int32)
            seed = 100 + k
            d2 = rng_actions(seed, len(envs), diff, A)
            for h in (fused, twin):
                h.step(ta)
                h.reset_done(3)         # (consumes the step's own list: nobody is final yet)
                h.done[cls == k] = 1    # the caller ends these episodes
                h.reset_done(seed)      # (flags written by the caller: this one compacts)
            for e, o in enumerate(envs):
                o.step(int(acts[e]), 0)
                if e % 64 == k:
                    o.reset_with(d2[:, e])
                    resets += 1
    for t in range(3 * cfg["depth_slope"] * diff + 5 if diff <= 5 else 90):
        seed = 900 + 7 * t
        acts = rng.integers(0, A, size=B)
        if t % 9 == 4:
            acts[::5] = A + 1  # "no gate": still uses depth
        done = twin.done.cpu().numpy().astype(bool)
        ta = torch.as_tensor(acts, device="cuda", dtype=torch.int32)
        fused.reset_done_step(seed, ta)
        twin.reset_done(seed)
        twin.step(ta)
        d2 = rng_actions(seed, len(envs), diff, A)
        for e, o in enumerate(envs):
            if done[e]:
                o.reset_with(d2[:, e])
                resets += 1
            o.step(int(acts[e]), 0)
        fused.sync()
        twin.sync()
        assert torch.equal(fused.reward.view(torch.int32), twin.reward.view(torch.int32)), t
        assert torch.equal(fused.done, twin.done) and torch.equal(fused.success, twin.success) and torch.equal(fused.depth, twin.depth), t
        ne = len(envs)
        np.testing.assert_array_equal(f32_bits(fused.reward.cpu().numpy()[:ne]), np.array([o.reward_bits() for o in envs], dtype=np.uint32), err_msg=f"t={t}")
        np.testing.assert_array_equal(fused.depth.cpu().numpy()[:ne], [o.depth() for o in envs])
    assert resets > 1.9 * len(envs)  # every env went through episode ends
    assert torch.equal(fused.get_state("packed"), twin.get_state("packed"))
    np.testing.assert_array_equal(fused.get_state("i64").cpu().numpy()[: len(envs)], np.stack([o.get_state() for o in envs]))
    if track:
        for e in (0, 3, len(envs) - 1):
            assert fused.solution(e) == envs[e].solution() == twin.solution(e)
    if dense:
        assert torch.equal(dt, fused.observe())
        np.testing.assert_array_equal(dt.cpu().numpy()[: len(envs)].reshape(len(envs), -1), np.stack([o.dense_obs() for o in envs]).reshape(len(envs), -1))


@pytest.mark.parametrize("n,B,diff,track,given_coins", [(16, 4096, 3, True, True), (16, 2048, 70, True, True), (9, 777, 4, False, True),
                                                         (16, 4096, 3, True, False), (12, 8192, 70, False, False),
                                                         (16, 1000, 1, False, True), (5, 8192, 64, True, True), (16, 65536, 100, False, False),
                                                         (24, 4096, 70, True, True), (20, 777, 3, True, True), (32, 2048, 66, False, True), (24, 1000, 1, True, True),
                                                         (28, 8192, 70, True, False)])
def test_reset_done_step_with_the_reference_default_options_equals_the_two_calls(n, B, diff, track, given_coins):
    """CliffordEnv with add_inverts: qg_vec_reset_done_step -- up to 16 qubits one launch (qm_reset_inv2_step_kernel: the step on two lanes per env; a reset env's first
    step by the tree's wave on the rows it holds -- gate, solution-log entry, the coin's inversion as ballots -- for long scrambles, on two lanes that read
    the fresh episode back behind the 16-lane and per-lane resets: short scrambles, and most of the batch finishing at once: difficulty 1) -- against
    reset_done + step on a twin and, with the coins given, against the oracle; with the handle's counter-RNG coins against the twin alone.  Beyond 16 qubits
    (64-bit rows) the two launches behind the one call: the one-launch form was built, passed these cases and was slower (EXPERIMENTS.md, round 5)."""
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=True, add_perms=False, track_solution=track, difficulty=diff, depth_slope=1 if diff > 20 else 2, max_depth=128)
    fused, twin = VecEnv("clifford", n, gs, B, **cfg), VecEnv("clifford", n, gs, B, **cfg)
    envs = [OracleEnv("clifford", n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(min(B, 300))] if given_coins else []
    seed0 = 5
    fused.reset(seed0)
    twin.reset(seed0)
    draws = rng_actions(seed0, max(len(envs), 1), diff, A)
    for e, o in enumerate(envs):
        o.reset_with(draws[:, e])
    rng = np.random.default_rng(B + diff)
    resets = 0
    steps = 3 * cfg["depth_slope"] * diff + 5 if diff <= 5 else 2 * diff + 20
    for t in range(steps):
        seed = 900 + 7 * t
        acts = rng.integers(0, A, size=B)
        if t % 9 == 4:
            acts[::5] = A + 1  # "no gate": still uses depth
        coins = rng.integers(0, 2, size=B).astype(np.uint8)
        done = twin.done.cpu().numpy().astype(bool)
        ta = torch.as_tensor(acts, device="cuda", dtype=torch.int64 if t % 2 else torch.int32)
        tc = torch.as_tensor(coins, device="cuda") if given_coins else None
        fused.reset_done_step(seed, ta, tc)
        twin.reset_done(seed)
        twin.step(ta, tc)
        if envs:
            d2 = rng_actions(seed, len(envs), diff, A)
            for e, o in enumerate(envs):
                if done[e]:
                    o.reset_with(d2[:, e])
                    resets += 1
                o.step(int(acts[e]), int(coins[e]))
        fused.sync()
        twin.sync()
        assert torch.equal(fused.reward.view(torch.int32), twin.reward.view(torch.int32)), t
        assert torch.equal(fused.done, twin.done) and torch.equal(fused.success, twin.success) and torch.equal(fused.depth, twin.depth), t
        if envs:
            ne = len(envs)
            np.testing.assert_array_equal(f32_bits(fused.reward.cpu().numpy()[:ne]), np.array([o.reward_bits() for o in envs], dtype=np.uint32), err_msg=f"t={t}")
            np.testing.assert_array_equal(fused.depth.cpu().numpy()[:ne], [o.depth() for o in envs])
    if envs:
        assert resets > len(envs)  # every env went through episode ends
        np.testing.assert_array_equal(fused.get_state("i64").cpu().numpy()[: len(envs)], np.stack([o.get_state() for o in envs]))
    assert torch.equal(fused.get_state("packed"), twin.get_state("packed"))
    assert torch.equal(fused.inverted, twin.inverted) if hasattr(fused, "inverted") else True
    if track:
        for e in (0, 3, min(B, 300) - 1):
            assert fused.solution(e) == twin.solution(e)
            if envs:
                assert fused.solution(e) == envs[e].solution()


@pytest.mark.parametrize("n,inverts", [(16, False), (24, False), (16, True)])  # (qm_reset_step_kernel; 64-bit rows: q64_reset_step_kernel; add_inverts: qm_reset_inv2_step_kernel)
def test_reset_done_step_inside_a_captured_graph_replays_like_eager_calls(n, inverts):
    """A graph of 16 x reset_done_step (its first call compacts the list from the flags, the others are single launches) replayed three times
    against the same calls made eagerly on a twin: the device-side lists and flag arrays of consecutive replays line up whatever the
    graph's length (here even and odd)."""
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset("clifford", n)
    A, B = len(gs), 8192
    for T in (16, 7):
        cfg = dict(add_inverts=inverts, add_perms=False, track_solution=inverts, difficulty=3, depth_slope=2, max_depth=128)
        g_env, twin = VecEnv("clifford", n, gs, B, **cfg), VecEnv("clifford", n, gs, B, **cfg)
        acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda")
        coins = torch.randint(0, 2, (T, B), dtype=torch.uint8, device="cuda") if inverts else [None] * T  # (given: a replay has the capture's step counter)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            g_env.reset(1)
            twin.reset(1)

            def body(env):
                for t in range(T):
                    env.reset_done_step(40 + t, acts[t], coins[t])

            body(g_env)  # eager pass on both
            body(twin)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                body(g_env)
            for _ in range(3):
                graph.replay()
                body(twin)
            torch.cuda.synchronize()
        g_env.sync()
        twin.sync()
        assert torch.equal(g_env.get_state("packed"), twin.get_state("packed")), T
        assert torch.equal(g_env.depth, twin.depth) and torch.equal(g_env.done, twin.done) and torch.equal(g_env.reward.view(torch.int32), twin.reward.view(torch.int32)), T


@pytest.mark.parametrize("kind,n,B,diff,frac,cfg", [
    ("linear_function", 8, 65536, 64, 0.01, dict(add_inverts=False, track_solution=False)),   # config 2's env, a collector's regime: 16 lanes per finished env
    ("linear_function", 8, 8192, 64, 0.03, dict(add_inverts=True, track_solution=True)),      # config 2 with the reference's defaults
    ("linear_function", 8, 1000, 64, 0.5, dict(add_inverts=False, track_solution=False)),     # full waves: the per-lane chain; ragged last wave
    ("linear_function", 5, 777, 37, 0.05, dict(add_inverts=True, track_solution=False)),      # draws do not divide by 16 or 4
    ("linear_function", 3, 4100, 7, 0.1, dict(add_inverts=False, track_solution=True)),       # fewer draws than lanes
    ("linear_function", 8, 300, 200, 0.12, dict(add_inverts=False, track_solution=False, metrics_weights={"n_layers": 0.05})),  # layer records
    ("permutation", 9, 65536, 16, 0.01, dict(add_inverts=False, track_solution=False)),       # config 1's env
    ("permutation", 16, 5000, 100, 0.04, dict(add_inverts=True, track_solution=True)),
    ("permutation", 4, 130, 9, 0.6, dict(add_inverts=True, track_solution=False)),
])
def test_word_layouts_reset_finished_envs_on_16_lanes_each_and_in_the_step_launch(kind, n, B, diff, frac, cfg):
    """LinearFunctionEnv <= 8 qubits / PermutationEnv <= 16 (one uint64 per env): qg_vec_reset_done composes a finished env's scramble on 16 lanes
    (GF(2) products / permutation composition of the lanes' runs of draws) and qg_vec_reset_done_step does that and the step of every env in
    one launch (word_reset_step_kernel).  Every env against the oracle (linear_function.rs:285-300, permutation.rs:175-192), finished envs chosen at
    random so that waves hold 0, 1, several or 64 of them; the two-call sequence on a twin handle must agree too."""
    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv

    side = int(round(n ** 0.5))
    gs = grid_gateset("permutation", side, side) if kind == "permutation" and side * side == n else line_gateset(kind, n)
    A = len(gs)
    full = dict(cfg, add_perms=False, difficulty=diff, depth_slope=1, max_depth=64)
    one, two = (VecEnv(kind, n, gs, B, seed=77, **full) for _ in range(2))
    ocfg = {k: (int(v) if isinstance(v, bool) else v) for k, v in full.items()}
    ov = OracleVec(OracleEnv(kind, n, gs, **ocfg), B)
    for g in (one, two):
        g.reset(3)
    ov.reset_seeded(3)
    gen = torch.Generator(device="cuda").manual_seed(8)
    rng = np.random.default_rng(1)
    ids = np.arange(B)
    from test_gpu_fullsize import _coins

    per_env = n if kind == "permutation" else n * n
    for t in range(10):
        fin = (rng.random(B) < frac).astype(np.uint8)
        if t == 3:
            fin[:] = 0        # nobody finished
        if t == 4:
            fin[:64] = 1      # a whole wave
        ftorch = torch.as_tensor(fin, device="cuda")
        natural = one.done.clone()
        for g in (one, two):
            g.done.copy_(torch.maximum(natural, ftorch))  # Env::reset for the chosen envs on top of the episodes that ended by themselves
        fin = np.maximum(fin, natural.cpu().numpy())
        acts = torch.randint(-1, A + 1, (B,), dtype=torch.int32, device="cuda", generator=gen)
        for g in (one, two):
            g.set_counters(50 + t, 0)
        one.reset_done_step(900 + t, acts)
        two.reset_done(900 + t)
        if t == 5:  # reset_done alone against the oracle, before the step
            ov.reset_seeded(900 + t, mask=fin)
            assert np.array_equal(two.get_state("i64").cpu().numpy(), ov.get_state(per_env)), "reset_done"
            assert np.array_equal(two.depth.cpu().numpy()[fin == 1], np.full(int(fin.sum()), min(diff, 64)))
        else:
            ov.reset_seeded(900 + t, mask=fin)
        two.step(acts)
        c = _coins(77, ids, 50 + t) if cfg.get("add_inverts") else None
        r, s, f, d = ov.step(acts.cpu().numpy(), c)
        for name, g in (("one launch", one), ("two calls", two)):
            g.sync()
            assert np.array_equal(f32_bits(g.reward.cpu().numpy()), f32_bits(r)), (name, t)
            assert np.array_equal(g.done.cpu().numpy(), f) and np.array_equal(g.success.cpu().numpy(), s), (name, t)
            assert np.array_equal(g.depth.cpu().numpy(), d), (name, t)
            assert np.array_equal(g.get_state("i64").cpu().numpy(), ov.get_state(per_env)), (name, t)
    if cfg.get("track_solution"):
        o_sol, o_len = ov.solutions(64)
        for g in (one, two):
            g_sol, g_len = g.solutions(64)
            assert np.array_equal(g_len, o_len) and np.array_equal(g_sol, o_sol)


@pytest.mark.parametrize("kind,n,B,inverts", [("clifford", 16, 64, False), ("clifford", 16, 100, True), ("clifford", 9, 257, False), ("linear_function", 20, 1000, False),
                                              ("clifford", 16, 20000, True), ("clifford", 16, 70000, False), ("clifford", 12, 200000, True),
                                              ("linear_function", 32, 300001, False), ("clifford", 24, 5000, True), ("clifford", 20, 66001, False),
                                              ("clifford", 32, 100, True), ("linear_function", 17, 37, False)])
def test_finishers_left_as_a_mask_at_assorted_batch_sizes(kind, n, B, inverts):
    """The step leaves its finishers as one bit per env and qg_vec_reset_done's workgroups count the mask themselves (device_common.hpp done_mask_*): batches
    below one wave, with ragged last waves and workgroups, and beyond 65 536 envs, where a thread's share of the mask is more than four words and the
    entry is found by a search over the partial sums.  step -> reset_done -> step ... with short episodes, EVERY env against the oracle after every
    step; reset_done_step (one launch where the handle has it) in between."""
    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv
    from test_gpu_fullsize import _coins

    gs = line_gateset(kind, n)
    A, diff = len(gs), 3
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=False, difficulty=diff, depth_slope=2, max_depth=128)
    gv = VecEnv(kind, n, gs, B, seed=31, **cfg)
    ov = OracleVec(OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}), B)
    gv.reset(2)
    ov.reset_seeded(2)
    gen = torch.Generator(device="cuda").manual_seed(B)
    ids = np.arange(B)
    resets = 0
    for t in range(14):
        acts = torch.randint(0, A, (B,), dtype=torch.int32, device="cuda", generator=gen)
        gv.set_counters(t, 0)
        if t % 3 == 2:
            fin = gv.done.cpu().numpy()
            gv.reset_done_step(700 + t, acts)
            ov.reset_seeded(700 + t, mask=fin)
            resets += int(fin.sum())
        else:
            gv.step(acts)
        r, s, f, d = ov.step(acts.cpu().numpy(), _coins(31, ids, t) if inverts else None)
        gv.sync()
        assert np.array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r)), t
        assert np.array_equal(gv.done.cpu().numpy(), f) and np.array_equal(gv.depth.cpu().numpy(), d), (t, np.nonzero(gv.done.cpu().numpy() != f)[0][:8])
        if t % 3 != 1:  # (before a reset_done_step the finishers stay for it)
            gv.reset_done(300 + t)
            ov.reset_seeded(300 + t, mask=f)
            resets += int(f.sum())
    assert resets > B
    assert np.array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense()), "final states"


@pytest.mark.parametrize("L,stagger", [(40, True), (3, False)])
def test_pauli_finishers_left_as_a_mask_by_the_step(L, stagger):
    """PauliGym 20q (compact layout) x 8 192 in a collector's loop: ptile_step1c_kernel<LIST> leaves its finishers as one bit per env, qg_vec_reset_done's
    workgroups count the mask themselves -- no compaction launch.  stagger: episodes of 40 steps whose ends are spread over time (1 / 40 of the batch
    per step: <= B / 32, a tree per finished env); else episodes of 3 steps, a third of the batch finishing in every step (the per-lane generator,
    its entries found by a search over the partial sums).  Every env against the oracle after every step: reward bits, is_final, and the
    observation (the regenerated targets) at the end."""
    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv

    n, B = 20, 8192
    gs = line_gateset("pauli", n)
    A = len(gs)
    cfg = dict(add_perms=False, track_solution=False, max_rotations=5, difficulty=64, pauli_diff_scale=8, depth_slope=1, max_depth=L)
    gv = VecEnv("pauli", n, gs, B, **cfg)
    ov = OracleVec(OracleEnv("pauli", n, gs, **{k: int(v) for k, v in cfg.items()}), B)
    gv.reset(5)
    ov.reset_seeded(5)
    gen = torch.Generator(device="cuda").manual_seed(L)
    ids = np.arange(B)
    all_env = torch.arange(B, device="cuda")

    def step_and_check(t):
        acts = torch.randint(0, A, (B,), dtype=torch.int32, device="cuda", generator=gen)
        gv.step(acts)
        r, s, f, d = ov.step(acts.cpu().numpy())
        gv.sync()
        assert np.array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r)), t
        assert np.array_equal(gv.done.cpu().numpy(), f), (t, np.nonzero(gv.done.cpu().numpy() != f)[0][:8])
        return f

    t = 0
    if stagger:
        for k in range(L):  # Env::reset for class k at time k (the caller raises the flags: these resets compact a list from them)
            f = step_and_check(t)
            t += 1
            gv.reset_done(100 + k)
            ov.reset_seeded(100 + k, mask=f)
            gv.done[all_env % L == k] = 1
            gv.reset_done(5000 + k)
            ov.reset_seeded(5000 + k, mask=(ids % L == k))
    resets = 0
    for k in range(L + 12 if stagger else 12):
        f = step_and_check(t)
        t += 1
        frac = f.mean()
        if stagger:
            assert frac <= 1.0 / 32, frac  # short enough for the trees
        gv.reset_done(9000 + k)  # (the step before left the mask)
        ov.reset_seeded(9000 + k, mask=f)
        resets += int(f.sum())
    assert resets > (B if stagger else 3 * B)
    gv.sync()
    assert np.array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense()), "states and regenerated targets"


@pytest.mark.parametrize("kind,n,inverts", [("clifford", 24, True), ("clifford", 20, False), ("linear_function", 32, False), ("clifford", 16, True)])
def test_staggered_finishers_left_as_a_mask_are_reset_by_trees(kind, n, inverts):
    """A collector's loop whose episodes (40 steps, 70 scramble gates) end spread over time: 1 / 40 of 8 192 envs per step -- <= B / 32, so
    qg_vec_reset_done runs a tree per finished env (the tree workgroups of q64_reset_done_kernel / qm_init_block), its entries found in the mask the
    step left (TILE64: the workgroups behind the trees count the same mask and find nothing to do).  Every env against the
    oracle after every step, and the states at the end."""
    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv
    from test_gpu_fullsize import _coins

    B, L, diff = 8192, 40, 70
    gs = line_gateset(kind, n)
    A = len(gs)
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=False, difficulty=diff, depth_slope=1, max_depth=L)
    gv = VecEnv(kind, n, gs, B, seed=77, **cfg)
    ov = OracleVec(OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}), B)
    gv.reset(5)
    ov.reset_seeded(5)
    gen = torch.Generator(device="cuda").manual_seed(n)
    ids = np.arange(B)
    all_env = torch.arange(B, device="cuda")
    t = 0

    fuse_with = None  # the seed of a reset_done that the next step takes along (qg_vec_reset_done_step: one launch where the handle has it)

    def step_and_check():
        nonlocal t, fuse_with
        acts = torch.randint(0, A, (B,), dtype=torch.int32, device="cuda", generator=gen)
        gv.set_counters(t, 0)
        if fuse_with is None:
            gv.step(acts)
        else:
            gv.reset_done_step(fuse_with, acts)
            fuse_with = None
        r, s, f, d = ov.step(acts.cpu().numpy(), _coins(77, ids, t) if inverts else None)
        gv.sync()
        assert np.array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r)), t
        assert np.array_equal(gv.done.cpu().numpy(), f) and np.array_equal(gv.depth.cpu().numpy(), d), (t, np.nonzero(gv.done.cpu().numpy() != f)[0][:8])
        t += 1
        return f

    for k in range(L):  # Env::reset for class k at time k (the caller raises the flags: these resets compact a list from them)
        f = step_and_check()
        gv.reset_done(100 + k)
        ov.reset_seeded(100 + k, mask=f)
        gv.done[all_env % L == k] = 1
        gv.reset_done(5000 + k)
        ov.reset_seeded(5000 + k, mask=(ids % L == k))
    resets = 0
    for k in range(L + 6):
        f = step_and_check()
        assert 0 < f.mean() <= 1.0 / 32, f.mean()  # short enough for the trees
        if k % 3 == 1:
            fuse_with = 9000 + k  # (reset_done + the next step as one call)
        else:
            gv.reset_done(9000 + k)  # (the step before left the mask)
        ov.reset_seeded(9000 + k, mask=f)
        resets += int(f.sum())
    assert resets > B
    if fuse_with is not None:
        gv.reset_done(fuse_with)
    assert np.array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense()), "final states"


@pytest.mark.parametrize("kind,n", [("pauli", 20), ("pauli", 12), ("clifford", 24), ("clifford", 16)])
def test_tree_grid_sized_from_a_short_list_walks_a_longer_one_in_rounds(kind, n):
    """The tree launch's grid follows the list lengths the handle's resets have reported (qgym_api.cpp reset_tree_grid): after a reset of 3 envs it is
    64 workgroups, and the next reset -- 250 of 8 192 envs, still a tree's list -- is walked in four rounds by them (PauliEnv: the scramble waves and the
    labels' wave meet again at every round's barriers, the products' hand-over words go on counting).  Both as lists compacted from raised flags and as
    masks left by a step; every env against the oracle."""
    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv

    B = 8192
    gs = line_gateset(kind, n)
    cfg = dict(add_perms=False, track_solution=False, difficulty=128, depth_slope=1, max_depth=200)
    if kind == "pauli":
        cfg.update(max_rotations=5, pauli_diff_scale=8)
    else:
        cfg.update(add_inverts=False)
    gv = VecEnv(kind, n, gs, B, **cfg)
    ov = OracleVec(OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}), B)
    gv.reset(2)
    ov.reset_seeded(2)
    rng = np.random.default_rng(n)
    for count, seed in ((3, 11), (250, 12), (5, 13), (256, 14), (200, 15)):
        ids = np.sort(rng.choice(B, size=count, replace=False))
        m = np.zeros(B, dtype=np.uint8)
        m[ids] = 1
        gv.done.copy_(torch.as_tensor(m, device="cuda"))
        gv.reset_done(seed)
        gv.sync()  # (the launch has reported its list's length)
        ov.reset_seeded(seed, mask=m)
        assert np.array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense()), (count, seed)
    # ... and with the finishers left as a MASK by a step: episodes of 40 steps whose ends are spread over time (205 envs per step), and between a
    # step's reset and the next step a reset of three hand-picked envs, which sizes the next tree launch down to 64 workgroups
    L = 40
    cfg2 = dict(cfg, difficulty=70, max_depth=L)
    g2 = VecEnv(kind, n, gs, B, **cfg2)
    o2 = OracleVec(OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg2.items()}), B)
    g2.reset(4)
    o2.reset_seeded(4)
    A = len(gs)
    gen = torch.Generator(device="cuda").manual_seed(n)
    ids, all_env = np.arange(B), torch.arange(B, device="cuda")
    t = 0

    def step_and_check():
        nonlocal t
        acts = torch.randint(0, A, (B,), dtype=torch.int32, device="cuda", generator=gen)
        g2.set_counters(t, 0)
        g2.step(acts)
        f = o2.step(acts.cpu().numpy())[2]
        g2.sync()
        assert np.array_equal(g2.done.cpu().numpy(), f), (t, np.nonzero(g2.done.cpu().numpy() != f)[0][:8])
        t += 1
        return f

    for k in range(L):  # Env::reset for class k at time k
        f = step_and_check()
        g2.reset_done(100 + k)
        o2.reset_seeded(100 + k, mask=f)
        g2.done[all_env % L == k] = 1
        g2.reset_done(5000 + k)
        o2.reset_seeded(5000 + k, mask=(ids % L == k))
    rounds = 0
    for k in range(12):
        f = step_and_check()
        assert 64 < f.sum() <= B // 32, f.sum()  # more than the sized-down grid, still a tree's list
        g2.reset_done(9000 + k)  # (the step before left the mask; the launch before reported 3 finishers)
        o2.reset_seeded(9000 + k, mask=f)
        few = np.zeros(B, dtype=np.uint8)
        few[rng.choice(B, size=3, replace=False)] = 1
        g2.done.copy_(torch.as_tensor(few, device="cuda"))
        g2.reset_done(9500 + k)
        g2.sync()
        o2.reset_seeded(9500 + k, mask=few)
        rounds += 1
        assert np.array_equal(g2.observe().cpu().numpy().reshape(B, -1), o2.observe_dense()), k
    assert rounds == 12


@pytest.mark.parametrize("case", range(16))
def test_reset_done_step_random_shapes_against_the_oracle(case):
    """Random env kind / size / options / batch / episode length: a collector's loop of qg_vec_reset_done_step calls -- one launch where the handle has it
    (32-bit rows plain and with add_inverts, 64-bit rows plain, the one-word layouts), the two launches behind the call elsewhere -- with EVERY env on the
    oracle after every step: reward bits, is_final, depth; states and solution logs at the end.  Episodes are short (they end, and end again, inside the
    loop); scrambles are short or long (the 16-lane / per-lane resets, or trees).  QGYM_FUZZ_SEED_OFFSET shifts the cases."""
    import os

    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv
    from test_gpu_fullsize import _coins

    rng = np.random.default_rng(777 + case + int(os.environ.get("QGYM_FUZZ_SEED_OFFSET", "0")))
    kind = ["clifford", "clifford", "linear_function", "permutation"][int(rng.integers(0, 4))]
    n = int(rng.integers(3, 33)) if kind == "clifford" else int(rng.integers(3, 41)) if kind == "linear_function" else int(rng.integers(3, 17))
    inverts = bool(rng.integers(0, 2))
    track = bool(rng.integers(0, 2))
    diff = int(rng.choice([1, 3, 7, 64, 70, 130]))
    L = int(rng.choice([2, 3, 5, 9]))
    B = int(rng.choice([70, 1000, 4097, 20000]))
    gs = line_gateset(kind, n)
    A = len(gs)
    seed0 = int(rng.integers(0, 1 << 30))
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=track, difficulty=diff, depth_slope=1, max_depth=L)
    gv = VecEnv(kind, n, gs, B, seed=seed0, **cfg)
    ov = OracleVec(OracleEnv(kind, n, gs, **{k: int(v) for k, v in cfg.items()}), B)
    gv.reset(11)
    ov.reset_seeded(11)
    gen = torch.Generator(device="cuda").manual_seed(case)
    ids = np.arange(B)
    what = f"{kind} {n}q inverts={inverts} track={track} diff={diff} L={L} B={B}"
    f = np.zeros(B, dtype=np.uint8)
    for t in range(3 * L + 4):
        acts = torch.randint(0, A + (1 if t % 5 == 4 else 0), (B,), dtype=torch.int32, device="cuda", generator=gen)  # (now and then an action past the gateset)
        gv.set_counters(t, 0)
        seed = 1000 + t
        if t == 0:
            gv.step(acts)
        else:
            gv.reset_done_step(seed, acts)
            ov.reset_seeded(seed, mask=f)
        r, s_, f, d = ov.step(acts.cpu().numpy(), _coins(seed0, ids, t) if inverts else None)
        gv.sync()
        assert np.array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r)), (what, t)
        assert np.array_equal(gv.done.cpu().numpy(), f) and np.array_equal(gv.depth.cpu().numpy(), d), (what, t, np.nonzero(gv.done.cpu().numpy() != f)[0][:8])
    assert np.array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense()), (what, "final states")
    if track:
        o_sol, o_len = ov.solutions(64)
        g_sol, g_len = gv.solutions(64)
        assert np.array_equal(g_len, o_len) and np.array_equal(g_sol, o_sol), what


@pytest.mark.parametrize("n,inverts", [(16, True), (16, False), (24, False)])
def test_bursts_of_finishers_inside_the_one_launch_pair(n, inverts):
    """The one-launch pair is chosen from the list lengths the handle has seen (qg_vec_reset_done_step: it pays where the resets are trees); a list that is
    suddenly LONG is then reset inside the fused kernel by its other roles: 6 000 of 262 144 envs finishing in one step after steps in which nobody finished
    (the 16-lane resets; with add_inverts their first steps on lane pairs), then the other 256 144 at once (the lane-per-env resets: 32 envs' first steps at a
    time on lane pairs).  Every env against the oracle after every step."""
    from oracle import OracleVec
    from qiskit_gym_amd.vec import VecEnv
    from test_gpu_fullsize import _coins

    B, K, L, diff = 262144, 6000, 8, 64
    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=inverts, difficulty=diff, depth_slope=1, max_depth=L)
    gv = VecEnv("clifford", n, gs, B, seed=99, **cfg)
    ov = OracleVec(OracleEnv("clifford", n, gs, **{k: int(v) for k, v in cfg.items()}), B)
    gv.reset(3)
    ov.reset_seeded(3)
    gen = torch.Generator(device="cuda").manual_seed(n)
    ids = np.arange(B)
    late = np.ones(B, dtype=np.uint8)
    late[np.random.default_rng(5).choice(B, size=K, replace=False)] = 0  # everybody but K envs starts again three steps in
    f = np.zeros(B, dtype=np.uint8)
    bursts = []
    for t in range(2 * L + 8):
        acts = torch.randint(0, A, (B,), dtype=torch.int32, device="cuda", generator=gen)
        gv.set_counters(t, 0)
        seed = 2000 + t
        if t == 0:
            gv.step(acts)
        else:
            gv.reset_done_step(seed, acts)
            ov.reset_seeded(seed, mask=f)
        r, s_, f, d = ov.step(acts.cpu().numpy(), _coins(99, ids, t) if inverts else None)
        gv.sync()
        assert np.array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r)), t
        assert np.array_equal(gv.done.cpu().numpy(), f) and np.array_equal(gv.depth.cpu().numpy(), d), (t, np.nonzero(gv.done.cpu().numpy() != f)[0][:8])
        bursts.append(int(f.sum()))
        if t == 2:  # (a reset from raised flags, a launch of its own: the K others are now three steps ahead)
            gv.reset_done(seed + 7)  # nobody is final: consumes the step's mask
            gv.done.copy_(torch.as_tensor(late, device="cuda"))
            gv.reset_done(777)
            ov.reset_seeded(777, mask=late)
            f = np.zeros(B, dtype=np.uint8)
            gv.sync()
    assert any(K // 2 < b <= K for b in bursts) and any(b > B // 2 for b in bursts), bursts  # both bursts happened (a few envs are solved early)
    assert np.array_equal(gv.observe().cpu().numpy().reshape(B, -1), ov.observe_dense()), "final states"
