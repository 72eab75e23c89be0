"""The bit-consuming first policy layer (qg_vec_embed) against a dense reference computed from
Env::observe.  Integer data pins the k permutation of the packed weights exactly; random data checks
the f32 accumulation within bf16 output rounding (tolerance stated below)."""
import numpy as np
import pytest
import torch

from qiskit_gym_amd import _lib
from qiskit_gym_amd.collector import embed, pack_embedding
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

pytestmark = pytest.mark.gpu

CASES = [  # kind, qubits, batch, hidden
    ("clifford", 16, 1000, 128),
    ("clifford", 16, 4096, 512),
    ("clifford", 5, 700, 64),
    ("clifford", 9, 63, 192),
    ("linear_function", 12, 1, 64),
    ("linear_function", 32, 1500, 128),
    ("linear_function", 17, 513, 64),
    ("linear_function", 27, 300, 64),   # 7 row groups (padded to 8 in the slab)
    ("clifford", 3, 200, 64),           # 2 row groups
    ("clifford", 16, 2048, 1024),       # 16 column slabs
    # up to 4 x CUs (env tile, slab) pairs the launch takes embed_small_kernel; beyond, embed_bits_kernel's 512-env passes
    ("clifford", 16, 9000, 128),
    ("linear_function", 27, 20001, 64),
    ("clifford", 5, 40000, 64),
    ("clifford", 16, 4097, 512),
]


def _env(kind, n, B, seed):
    gs = line_gateset(kind, n)
    env = VecEnv(kind, n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=64)
    env.reset(seed)
    return env


@pytest.mark.parametrize("kind,n,B,hidden", CASES)
def test_embed_integer_weights_exact(kind, n, B, hidden):
    env = _env(kind, n, B, 11)
    obs = env.observe().to(torch.float64).flatten(1)  # [B, D*D]
    K = obs.shape[1]
    g = torch.Generator(device="cpu").manual_seed(5)
    # +-1 entries, at most 200 per output: every partial sum is an integer below 256, exact in bf16
    w = torch.zeros((hidden, K), dtype=torch.float32)
    for r in range(hidden):
        idx = torch.randperm(K, generator=g)[: min(K, 200)]
        w[r, idx] = (torch.randint(0, 2, (idx.numel(),), generator=g) * 2 - 1).float()
    bias = torch.randint(-8, 9, (hidden,), generator=g).float()
    wd, bd = w.cuda(), bias.cuda()
    for wt in (wd, wd.to(torch.bfloat16)):
        packed = pack_embedding(env, wt)
        for relu in (False, True):
            out = embed(env, packed, bd, hidden, relu=relu)
            ref = obs @ wd.double().t() + bd.double()
            if relu:
                ref = ref.clamp_min(0)
            assert torch.equal(out.double(), ref), f"{kind} {n}q: mismatch in {(out.double() != ref).sum().item()} entries"
    out = embed(env, packed, None, hidden, relu=False)
    assert torch.equal(out.double(), obs @ wd.double().t())
    env.sync()


@pytest.mark.parametrize("kind,n,B,hidden", CASES[:2] + CASES[5:6])
def test_embed_random_weights(kind, n, B, hidden):
    env = _env(kind, n, B, 3)
    obs = env.observe().to(torch.float64).flatten(1)
    K = obs.shape[1]
    g = torch.Generator(device="cpu").manual_seed(9)
    w = (torch.randn((hidden, K), generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(hidden, generator=g).cuda()
    out = embed(env, pack_embedding(env, w), bias, hidden, relu=True)
    ref = (obs @ w.double().t() + bias.double()).clamp_min(0)
    # f32 accumulation of exact products (|error| <~ 1e-6 * sum|w|) then one bf16 rounding (2^-9 relative)
    err = (out.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0**-8 + 1e-4).all()), f"max error {err.max().item()}"
    # a strided output (the layer may write into a wider activation buffer)
    wide = torch.zeros((B, hidden + 64), dtype=torch.bfloat16, device="cuda")
    embed(env, pack_embedding(env, w), bias, hidden, relu=True, out=wide[:, :hidden])
    assert torch.equal(wide[:, :hidden], out) and not bool(wide[:, hidden:].any())
    env.sync()


@pytest.mark.parametrize("kind,n,B", [("clifford", 16, 1000), ("clifford", 5, 700), ("clifford", 12, 33), ("clifford", 9, 4096), ("linear_function", 12, 64),
                                      ("linear_function", 27, 300), ("linear_function", 32, 1500), ("clifford", 16, 9000), ("linear_function", 17, 20001)])
def test_embed_observe_also_writes_the_packed_observation(kind, n, B):
    """qg_vec_embed_observe = observe_packed + embed of the same state (small batches: the layer's kernel writes both)."""
    env = _env(kind, n, B, 21)
    K = env.obs_shape_[0] * env.obs_shape_[1]
    w = torch.randint(-1, 2, (64, K), generator=torch.Generator().manual_seed(4)).float()
    w[:, 150:] = 0  # few nonzeros per output: exact in bf16
    packed = pack_embedding(env, w.cuda())
    want_obs = env.observe_packed()
    want = embed(env, packed, None, 64, relu=False)
    obs = torch.full_like(want_obs, -1)
    got = embed(env, packed, None, 64, relu=False, obs_out=obs)
    assert torch.equal(got, want)
    assert torch.equal(obs, want_obs)
    with pytest.raises(ValueError):
        embed(env, packed, None, 64, relu=False, obs_out=obs[:, :-1])
    env.sync()


@pytest.mark.parametrize("n,B", [(12, 1000), (16, 333), (27, 64), (32, 2500), (9, 9000), (20, 20001)])
def test_embed_reads_the_dual_layout_of_linear_function_envs_with_inverts(n, B):
    """LinearFunctionEnv with add_inverts (the reference default) keeps the matrix and its inverse side by side and a flag per env says
    which is the state (kernels_lfd.hip): the first layer and the packed observation follow the flag, env by env."""
    gs = line_gateset("linear_function", n)
    env = VecEnv("linear_function", n, gs, B, add_inverts=True, add_perms=False, track_solution=False, difficulty=40)
    env.reset(7)
    g = torch.Generator(device="cuda").manual_seed(n)
    for t in range(5):  # coins flip the flag of about half the envs per step
        env.rollout(torch.randint(0, len(gs), (1, B), device="cuda", generator=g, dtype=torch.int32),
                    coins=torch.randint(0, 2, (1, B), device="cuda", generator=g, dtype=torch.uint8))
    K = n * n
    w = torch.randint(-1, 2, (128, K), generator=torch.Generator().manual_seed(4)).float()
    w[:, 180:] = 0  # few nonzeros per output: exact in bf16
    bias = torch.randint(-8, 9, (128,), generator=torch.Generator().manual_seed(5)).float().cuda()
    packed = pack_embedding(env, w.cuda())
    ref = (env.observe().double().flatten(1) @ w.cuda().double().t() + bias.double()).clamp_min(0)
    want_obs = env.observe_packed()
    obs = torch.full_like(want_obs, -1)
    got = embed(env, packed, bias, 128, relu=True, obs_out=obs)
    assert torch.equal(got.double(), ref)
    assert torch.equal(obs, want_obs)
    assert torch.equal(embed(env, packed, bias, 128, relu=True).double(), ref)
    env.sync()


def test_embed_follows_the_state():
    """The layer reads the live tiles: after a step it sees the new observation."""
    env = _env("clifford", 16, 2048, 1)
    w = torch.randint(-1, 2, (64, 1024), generator=torch.Generator().manual_seed(2)).float()
    w[:, 200:] = 0  # few nonzeros per output: exact in bf16
    w = w.cuda()
    packed = pack_embedding(env, w)
    for t in range(4):
        acts = torch.randint(0, env.num_actions(), (2048,), device="cuda", dtype=torch.int32)
        env.step(acts)
        ref = env.observe().double().flatten(1) @ w.double().t()
        assert torch.equal(embed(env, packed, None, 64, relu=False).double(), ref)
    env.sync()


def test_embed_unsupported_layouts():
    gs = line_gateset("clifford", 20)
    env = VecEnv("clifford", 20, gs, 64, add_inverts=False, add_perms=False, track_solution=False)
    with pytest.raises((ValueError, _lib.QGymError)):
        pack_embedding(env, torch.zeros((64, 1600), device="cuda"))
    env16 = _env("clifford", 16, 64, 1)
    with pytest.raises((ValueError, _lib.QGymError)):
        pack_embedding(env16, torch.zeros((100, 1024), device="cuda"))  # hidden % 64 != 0
