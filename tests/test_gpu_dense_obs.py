"""The dense int8 observation the Gym adapter returns after every step (reference src/qiskit_gym/envs/adapters.py:50-54,62-72;
Clifford::observe rust/src/envs/clifford.rs:361-368): the streaming full rewrite (qg_vec_observe_dense) and the resident, incrementally
maintained form (qg_vec_track_dense) against the CPU oracle's observe(), bit for bit, across every call that changes states."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import OracleEnv  # noqa: E402
from util import f32_bits, line_gateset, make_pair, rng_actions  # noqa: E402


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda").to(dtype)


TRACKED = [("clifford", 16), ("clifford", 8), ("linear_function", 16), ("linear_function", 32)]
ANY_SIZE = [("clifford", 3), ("clifford", 5), ("clifford", 12), ("clifford", 15), ("linear_function", 9), ("linear_function", 13), ("linear_function", 31)]


@pytest.mark.parametrize("kind,n", TRACKED + ANY_SIZE)
@pytest.mark.parametrize("batch", [1, 63, 64, 1000])
def test_streaming_dense_observation_matches_oracle(kind, n, batch):
    """qg_vec_observe_dense on the shapes the streaming kernel serves, ragged last tile included."""
    gs = line_gateset(kind, n)
    ov, gv = make_pair(kind, n, gs, batch, add_inverts=False, add_perms=False, track_solution=False, difficulty=3 * n)
    rng = np.random.default_rng(n + batch)
    draws = rng.integers(0, len(gs), size=(3 * n, batch))
    ov.reset_with(draws)
    gv.reset_with(_dev(draws, torch.int32))
    guard = torch.full((batch * gv.obs_shape_[0] * gv.obs_shape_[1] + 64,), 7, dtype=torch.int8, device="cuda")
    out = guard[: batch * gv.obs_shape_[0] * gv.obs_shape_[1]].view(batch, *gv.obs_shape_)
    gv.observe(out=out)
    gv.sync()
    np.testing.assert_array_equal(out.cpu().numpy().reshape(batch, -1), ov.observe_dense())
    assert (guard[-64:] == 7).all(), "the kernel wrote past the last env"


@pytest.mark.parametrize("kind,n", TRACKED)
@pytest.mark.parametrize("track_solution", [False, True])
def test_tracked_dense_observation_follows_every_step(kind, n, track_solution):
    """track_dense(): after reset_with, single steps (in-range, out-of-range and negative actions), a graph rollout, a fused rollout and
    set_state the resident tensor equals the oracle's dense observation -- and a fresh observe()."""
    gs = line_gateset(kind, n)
    A, B = len(gs), 517
    ov, gv = make_pair(kind, n, gs, B, add_inverts=False, add_perms=False, track_solution=track_solution, difficulty=2 * n, max_depth=200)
    rng = np.random.default_rng(7 * n)
    dense = gv.track_dense()

    def check(label):
        gv.sync()
        np.testing.assert_array_equal(dense.cpu().numpy().reshape(B, -1), ov.observe_dense(), err_msg=label)
        assert torch.equal(dense, gv.observe()), label

    check("constructor state")
    draws = rng.integers(0, A, size=(2 * n, B))
    ov.reset_with(draws)
    gv.reset_with(_dev(draws, torch.int32))
    check("after reset_with")
    for t in range(24):
        acts = rng.integers(0, A, size=B)
        if t % 4 == 3:
            acts[::7] = A + 2
            acts[1::9] = -1
        r, s, f, d = ov.step(acts)
        gv.step(_dev(acts, torch.int32))
        check(f"step {t}")
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), f32_bits(r))
        np.testing.assert_array_equal(gv.success.cpu().numpy(), s)
    for fused in (False, True):  # T single-step launches of one graph (tracked in-kernel) / one fused launch (followed by a full rewrite)
        T = 6
        seq = rng.integers(0, A, size=(T, B))
        for t in range(T):
            ov.step(seq[t])
        gv.rollout(_dev(seq, torch.int32), fused=fused)
        check(f"rollout fused={fused}")
    st = ov.get_state(gv.obs_shape_[0] * gv.obs_shape_[1])
    perm = rng.permutation(B)
    ov.set_state(st[perm])
    gv.set_state(st[perm], "i64")
    check("after set_state")
    gv.track_dense(False)  # detached: later steps leave the tensor alone
    before = dense.clone()
    acts = rng.integers(0, A, size=B)
    gv.step(_dev(acts, torch.int32))
    gv.sync()
    assert torch.equal(dense, before)


@pytest.mark.parametrize("n,batch", [(16, 517), (16, 64), (8, 300)])
def test_tracked_dense_observation_with_the_reference_default_options(n, batch):
    """add_inverts=True, track_solution=True (envs/synthesis.py:182-204): CliffordEnv 16q's two-lanes-per-env step rewrites an env's whole
    observation when its coin inverts the matrix and the gate's rows otherwise; 8 qubits: a full rewrite after every step.  Single steps with
    given coins, a graph rollout, a fused rollout (Gauss-Jordan-free thread-per-env kernel + full rewrite), a non-symplectic set_state."""
    gs = line_gateset("clifford", n)
    A, B = len(gs), batch
    ov, gv = make_pair("clifford", n, gs, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=2 * n, max_depth=200)
    rng = np.random.default_rng(n * B)
    dense = gv.track_dense()

    def check(label):
        gv.sync()
        np.testing.assert_array_equal(dense.cpu().numpy().reshape(B, -1), ov.observe_dense(), err_msg=label)

    draws = rng.integers(0, A, size=(2 * n, B))
    ov.reset_with(draws)
    gv.reset_with(_dev(draws, torch.int32))
    check("after reset_with")
    for t in range(20):
        acts = rng.integers(0, A, size=B)
        coins = rng.integers(0, 2, size=B) if t % 5 else np.full(B, t % 2)  # all inverted / none inverted now and then
        if t == 7:
            acts[::6] = A + 1  # "no gate": nothing but the inversion changes the matrix
        ov.step(acts, coins)
        gv.step(_dev(acts, torch.int32), _dev(coins, torch.uint8))
        check(f"step {t}")
    for fused in (False, True):
        T = 5
        seq, cseq = rng.integers(0, A, size=(T, B)), rng.integers(0, 2, size=(T, B))
        for t in range(T):
            ov.step(seq[t], cseq[t])
        gv.rollout(_dev(seq, torch.int32), fused=fused, coins=_dev(cseq, torch.uint8))
        check(f"rollout fused={fused}")
    # an arbitrary invertible (not symplectic) matrix: the Gauss-Jordan step kernel runs, followed by a full rewrite
    st = ov.get_state(4 * n * n)
    st[:, : 2 * n] = 0
    st[:, 0] = 1
    st[:, 1] = 1  # row 0 := e0 + e1: invertible with the other rows? keep it simple: only env 0 gets it, if its row 1 is not equal
    ov.set_state(st)
    gv.set_state(st, "i64")
    check("after set_state")


@pytest.mark.parametrize("inverts", [False, True])
def test_tracked_dense_observation_across_auto_reset(inverts):
    """Episodes that end are re-scrambled by reset_done (a list of finished envs: the reset rewrites their observations itself; with
    add_inverts the two-lanes-per-env step is followed by a full rewrite): the resident tensor stays equal to the oracle's."""
    from qiskit_gym_amd.vec import VecEnv

    n, B, diff = 16, 700, 3
    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=False, difficulty=diff, depth_slope=2, max_depth=128)
    gv = VecEnv("clifford", n, gs, B, **cfg)
    envs = [OracleEnv("clifford", n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(B)]
    dense = gv.track_dense()
    gv.reset(21)
    draws = rng_actions(21, B, diff, A)
    for e, o in enumerate(envs):
        o.reset_with(draws[:, e])
    rng = np.random.default_rng(5)
    resets = 0
    for t in range(20):
        done = gv.done.cpu().numpy().astype(bool)
        if done.any():
            seed = 500 + t
            gv.reset_done(seed)
            d2 = rng_actions(seed, B, diff, A)
            for e in np.nonzero(done)[0]:
                envs[e].reset_with(d2[:, e])
                resets += 1
        acts = rng.integers(0, A, size=B)
        coins = rng.integers(0, 2, size=B)
        for o, a, c in zip(envs, acts, coins):
            o.step(int(a), int(c))
        gv.step(_dev(acts, torch.int32), _dev(coins, torch.uint8))
        gv.sync()
        want = np.stack([o.dense_obs() for o in envs]).reshape(B, -1)
        np.testing.assert_array_equal(dense.cpu().numpy().reshape(B, -1), want, err_msg=f"t={t}")
    assert resets > B


def test_track_dense_refuses_unsupported_shapes():
    from qiskit_gym_amd.vec import VecEnv

    for kind, n in (("clifford", 5), ("linear_function", 8), ("clifford", 20), ("permutation", 9)):
        gv = VecEnv(kind, n, line_gateset(kind, n), 64, add_inverts=False)
        with pytest.raises(Exception, match="track_dense"):
            gv.track_dense()


def test_tracked_dense_at_full_size_in_a_graph():
    """The bench leg's shape: 65 536 envs, one captured graph of single-step launches that keep the dense observation current; the
    tensor is compared with a full rewrite of the same state and, on a sample, with the oracle."""
    from qiskit_gym_amd.vec import VecEnv
    from oracle import OracleVec

    n, B, T = 16, 65536, 32
    gs = line_gateset("clifford", n)
    A = len(gs)
    gv = VecEnv("clifford", n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=64)
    dense = gv.track_dense()
    gv.reset(77)
    acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda")
    gv.rollout(acts)  # builds the graph
    gv.rollout(acts)  # replays it
    gv.sync()
    assert torch.equal(dense, gv.observe())
    ids = np.arange(0, B, 257)
    proto = OracleEnv("clifford", n, gs, add_inverts=0, add_perms=0, track_solution=0, difficulty=64)
    ov = OracleVec(proto, len(ids))
    ov.reset_with(rng_actions(77, ids, 64, A))
    host = acts.cpu().numpy()[:, ids]
    for _ in range(2):
        for t in range(T):
            ov.step(host[t])
    np.testing.assert_array_equal(dense[torch.as_tensor(ids, device="cuda")].cpu().numpy().reshape(len(ids), -1), ov.observe_dense())
