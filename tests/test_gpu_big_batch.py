"""An env's trajectory depends on its global id only (seed, env_base + index, step): a handle of 2^20 + 77 envs and the same envs
held by three smaller handles with env_base offsets -- the multi-GPU partitioning on one card -- agree bit for bit, and a sample of
them (first, middle, last, strided) agrees with the oracle, out-of-range actions included."""
import numpy as np
import pytest
import torch

from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,n", [("clifford", 16), ("clifford", 7), ("linear_function", 20)])
def test_one_big_handle_equals_its_env_base_parts(kind, n):
    B = (1 << 20) + 77  # a ragged last tile
    gs = line_gateset(kind, n)
    kw = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=24, seed=5)
    big = VecEnv(kind, n, gs, B, **kw)
    cuts = [0, 1 << 19, 1 << 20, B]
    parts = [VecEnv(kind, n, gs, cuts[i + 1] - cuts[i], env_base=cuts[i], **kw) for i in range(3)]
    big.reset(3)
    for p in parts:
        p.reset(3)
    A = len(gs)
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(5):
        acts = torch.randint(-1, A + 1, (B,), device="cuda", generator=g, dtype=torch.int32)  # with out-of-range actions on both sides
        r, d = big.step(acts)
        obs = big.observe_packed()
        for i, p in enumerate(parts):
            sl = slice(cuts[i], cuts[i + 1])
            pr, pd = p.step(acts[sl].contiguous())
            assert torch.equal(pr, r[sl]) and torch.equal(pd, d[sl])
            assert torch.equal(p.observe_packed(), obs[sl])
    big.sync()


def test_odd_million_env_batch_against_the_oracle():
    from oracle import OracleEnv, OracleVec
    from util import f32_bits, rng_actions

    n, B, T, diff = 16, (1 << 20) + 77, 6, 24
    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    gv = VecEnv("clifford", n, gs, B, **cfg)
    gv.reset(31)
    half = B // 2
    ids = np.concatenate([np.arange(0, 70), np.arange(half - 70, half + 70), np.arange(B - 140, B), np.arange(4321, B, 40009)])
    ov = OracleVec(OracleEnv("clifford", n, gs, **cfg), len(ids))
    ov.reset_with(rng_actions(31, ids, diff, A))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(6)
    idx = torch.as_tensor(ids, device="cuda")
    for t in range(T):
        acts = torch.randint(-1, A + 1, (B,), dtype=torch.int32, device="cuda", generator=gen)  # with out-of-range actions
        gv.step(acts)
        r, s, f, d = ov.step(acts[idx].cpu().numpy())
        np.testing.assert_array_equal(f32_bits(gv.reward[idx].cpu().numpy()), f32_bits(r))
        np.testing.assert_array_equal(gv.depth[idx].cpu().numpy(), d)
        np.testing.assert_array_equal(gv.done[idx].cpu().numpy(), f)
        np.testing.assert_array_equal(gv.success[idx].cpu().numpy(), s)
    want = (ov.observe_dense().reshape(len(ids), 32, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)
    np.testing.assert_array_equal(gv.observe_packed()[idx].cpu().numpy().view(np.uint32), want)
    gv.sync()
