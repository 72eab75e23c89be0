"""Collector kernels (kernels_collect.hip) against their numpy restatements: observation expansion
(bit-exact), action sampling (exact winner of the same exponential race, log-prob / entropy to f32
round-off, distribution test), GAE (bit-exact f32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from collect_ref import gae_f32, log_softmax, race_keys, sample_uniforms  # noqa: E402
from util import f32_bits, grid_gateset, line_gateset  # noqa: E402

DTYPES = [torch.int8, torch.bfloat16, torch.float16, torch.float32]


@pytest.mark.parametrize("word,cols", [(4, 32), (4, 16), (4, 7), (8, 64), (8, 40), (8, 33), (1, 9), (1, 16)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_expand_packed_matches_numpy(word, cols, dtype):
    from qiskit_gym_amd.collector import expand_packed

    rng = np.random.default_rng(word * 100 + cols)
    n_rows = 1000 * 3 + 5
    if word == 1:
        packed = rng.integers(0, cols, size=n_rows, dtype=np.uint8)
        want = (packed[:, None] == np.arange(cols)[None, :]).astype(np.float32)
    else:
        raw = rng.integers(0, 2**63, size=n_rows, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n_rows, dtype=np.uint64)
        packed = raw.astype(np.uint32 if word == 4 else np.uint64)
        want = ((packed[:, None].astype(np.uint64) >> np.arange(cols, dtype=np.uint64)[None, :]) & 1).astype(np.float32)
    tdt = {1: torch.uint8, 4: torch.int32, 8: torch.int64}[word]
    t = torch.from_numpy(packed.view({1: np.uint8, 4: np.int32, 8: np.int64}[word])).to("cuda").view(tdt)
    got = expand_packed(t, cols, dtype)
    assert got.shape == (n_rows, cols) and got.dtype == dtype
    np.testing.assert_array_equal(got.float().cpu().numpy(), want)


@pytest.mark.parametrize("kind,n", [("clifford", 16), ("clifford", 5), ("clifford", 20), ("linear_function", 8), ("linear_function", 12),
                                    ("linear_function", 40), ("permutation", 9), ("pauli", 6)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_observe_as_equals_the_int8_observation(kind, n, dtype):
    from qiskit_gym_amd.vec import VecEnv

    gs = grid_gateset("permutation", 3, 3) if kind == "permutation" else line_gateset(kind, n)
    B = 517
    cfg = dict(add_perms=False, track_solution=False, difficulty=7)
    if kind != "pauli":
        cfg["add_inverts"] = False
    env = VecEnv(kind, n, gs, B, **cfg)
    env.reset(3)
    acts = torch.randint(0, len(gs), (5, B), dtype=torch.int32, device="cuda")
    env.rollout(acts)
    want = env.observe().reshape(B, -1).float()
    got = env.observe_as(dtype)
    env.sync()
    assert got.dtype == dtype and got.shape == want.shape
    assert torch.equal(got.float(), want)
    assert set(np.unique(want.cpu().numpy())) <= {0.0, 1.0} and want.sum() > 0


@pytest.mark.parametrize("ldt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("A,pad", [(170, 6), (12, 0), (1, 0), (37, 3)])
def test_sample_actions_is_the_winner_of_the_exponential_race(ldt, A, pad):
    from qiskit_gym_amd.collector import sample_actions

    B, seed, counter = 3000, 1234, 17
    g = torch.Generator(device="cuda")
    g.manual_seed(A)
    full = (torch.randn((B, A + pad), device="cuda", generator=g) * 3).to(ldt)
    acts, logp, ent, vals = sample_actions(full, seed, counter, num_actions=A, value_col=(A if pad else None))
    torch.cuda.synchronize()
    logits = full[:, :A].float().cpu().numpy()
    keys = race_keys(logits, sample_uniforms(seed, B, counter, A))
    order = np.sort(keys, axis=1)
    margin = order[:, 1] - order[:, 0] if A > 1 else np.full(B, np.inf)
    want = keys.argmin(axis=1)
    got = acts.cpu().numpy()
    clear = margin > 1e-4  # f32 log vs f64 log may reorder keys that are this close (none or a handful)
    assert clear.mean() > 0.99
    np.testing.assert_array_equal(got[clear], want[clear])
    assert ((got >= 0) & (got < A)).all()
    lsm = log_softmax(logits)
    np.testing.assert_allclose(logp.cpu().numpy(), lsm[np.arange(B), got], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), -(np.exp(lsm) * lsm).sum(axis=1), rtol=0, atol=5e-5)
    if pad:
        np.testing.assert_array_equal(vals.cpu().numpy(), full[:, A].float().cpu().numpy())
    # a different counter gives different draws; the same one reproduces
    again, *_ = sample_actions(full, seed, counter, num_actions=A)
    other, *_ = sample_actions(full, seed, counter + 1, num_actions=A)
    assert torch.equal(again, acts)
    if A > 1:
        assert not torch.equal(other, acts)


def test_sample_actions_respects_masks_and_int32_output():
    from qiskit_gym_amd.collector import sample_actions

    B, A = 2048, 28
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    logits = torch.randn((B, A), device="cuda", generator=g)
    mask = (torch.rand((B, A), device="cuda", generator=g) < 0.4).to(torch.uint8)
    mask[:7] = 0  # finished envs: masks() is all false (clifford.rs:349-351)
    acts = torch.empty(B, dtype=torch.int32, device="cuda")
    _, logp, ent, _ = sample_actions(logits, 9, 0, mask=mask, actions=acts)
    torch.cuda.synchronize()
    a, m = acts.cpu().numpy(), mask.cpu().numpy().astype(bool)
    some = m.any(axis=1)
    assert m[np.arange(B), a][some].all()
    assert (a[~some] == 0).all() and (logp.cpu().numpy()[~some] == 0).all()
    lsm = log_softmax(logits.cpu().numpy(), m)
    np.testing.assert_allclose(logp.cpu().numpy()[some], lsm[np.arange(B), a][some], atol=2e-5, rtol=0)
    keys = race_keys(logits.cpu().numpy(), sample_uniforms(9, B, 0, A), m)
    srt = np.sort(keys, axis=1)
    with np.errstate(invalid="ignore"):
        clear = some & ((srt[:, 1] - srt[:, 0] > 1e-4) | ~np.isfinite(srt[:, 1]))
    np.testing.assert_array_equal(a[clear], keys.argmin(axis=1)[clear])


def test_sample_actions_follows_softmax():
    """200k draws from one 6-way distribution: chi-square against softmax(logits)."""
    from qiskit_gym_amd.collector import sample_actions

    B = 200_000
    row = torch.tensor([0.3, -1.2, 2.0, 0.0, 1.1, -3.0], device="cuda")
    logits = row.repeat(B, 1).contiguous()
    acts, *_ = sample_actions(logits, 42, 3)
    counts = np.bincount(acts.cpu().numpy(), minlength=6).astype(np.float64)
    p = torch.softmax(row.double(), 0).cpu().numpy()
    chi2 = ((counts - B * p) ** 2 / (B * p)).sum()
    assert chi2 < 30.0, (chi2, counts, B * p)  # 5 dof: P(chi2 > 30) ~ 1.5e-5


@pytest.mark.parametrize("T,B,with_last", [(33, 1000, True), (1, 70, True), (128, 257, False)])
def test_gae_is_bit_exact(T, B, with_last):
    from qiskit_gym_amd.collector import gae

    rng = np.random.default_rng(T)
    r = rng.normal(size=(T, B)).astype(np.float32)
    v = rng.normal(size=(T, B)).astype(np.float32)
    d = (rng.random((T, B)) < 0.1).astype(np.uint8)
    lv = rng.normal(size=B).astype(np.float32) if with_last else None
    adv, ret = gae(torch.from_numpy(r).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(d).cuda(),
                   torch.from_numpy(lv).cuda() if with_last else None, 0.995, 0.95)
    want_adv, want_ret = gae_f32(r, v, d, lv, 0.995, 0.95)
    np.testing.assert_array_equal(f32_bits(adv.cpu().numpy()), f32_bits(want_adv))
    np.testing.assert_array_equal(f32_bits(ret.cpu().numpy()), f32_bits(want_ret))


@pytest.mark.parametrize("A,K,B", [(170, 256, 3001), (12, 64, 500), (190, 128, 777), (33, 512, 64), (1, 64, 40), (214, 256, 1000), (222, 64, 90)])
def test_head_sample_is_the_race_on_its_own_logits(A, K, B):
    """qg_policy_head_sample = last layer + qg_sample_actions in one kernel: the action is the winner of the same
    exponential race on logits h W^T + b (f64 reference; bias carried as two bf16 terms), and log-prob / entropy /
    value are those of that row."""
    from qiskit_gym_amd.collector import head_sample, pack_head

    seed, counter = 99, 5
    g = torch.Generator(device="cuda")
    g.manual_seed(A * 1000 + K)
    h = (torch.randn((B, K), device="cuda", generator=g)).clamp_min(0).to(torch.bfloat16)
    w = (torch.randn((A + 3, K), device="cuda", generator=g) * (2.0 / K**0.5)).to(torch.bfloat16)
    b = torch.randn(A + 3, device="cuda", generator=g).to(torch.bfloat16)
    value_row = A + 1  # any row of the matrix may be the value head
    packed = pack_head(w, b, A, value_row)
    acts, logp, ent, vals = head_sample(h, packed, A, seed, counter)
    torch.cuda.synchronize()
    full = (h.double() @ w.double().t() + b.double()).cpu().numpy()
    logits = full[:, :A]
    keys = race_keys(logits, sample_uniforms(seed, B, counter, A))
    order = np.sort(keys, axis=1)
    margin = order[:, 1] - order[:, 0] if A > 1 else np.full(B, np.inf)
    got = acts.cpu().numpy()
    clear = margin > 1e-3  # f32 MFMA accumulation vs f64: keys this close may swap
    assert clear.mean() > 0.98
    np.testing.assert_array_equal(got[clear], keys.argmin(axis=1)[clear])
    assert ((got >= 0) & (got < A)).all()
    lsm = log_softmax(logits)
    np.testing.assert_allclose(logp.cpu().numpy(), lsm[np.arange(B), got], rtol=0, atol=2e-4)
    np.testing.assert_allclose(ent.cpu().numpy(), -(np.exp(lsm) * lsm).sum(axis=1), rtol=0, atol=2e-4)
    np.testing.assert_allclose(vals.cpu().numpy(), full[:, value_row], rtol=0, atol=2e-4)
    # f32 weights pack to the same thing as their bf16 rounding; int32 actions; reproducible; the counter matters
    a32 = torch.empty(B, dtype=torch.int32, device="cuda")
    head_sample(h, pack_head(w.float(), b.float(), A, value_row), A, seed, counter, actions=a32)
    assert torch.equal(a32.long(), acts)
    other, *_ = head_sample(h, packed, A, seed, counter + 1)
    if A > 1:
        assert not torch.equal(other, acts)


def test_head_sample_follows_softmax():
    """200k draws from one 6-way distribution through the fused head: chi-square against softmax."""
    from qiskit_gym_amd.collector import head_sample, pack_head

    B, K = 200_000, 64
    row = torch.tensor([0.3, -1.2, 2.0, 0.0, 1.1, -3.0], device="cuda")
    w = torch.zeros((7, K), device="cuda")
    w[:6, 0] = row  # h = e_0: logits = row exactly
    h = torch.zeros((B, K), dtype=torch.bfloat16, device="cuda")
    h[:, 0] = 1.0
    acts, *_ = head_sample(h, pack_head(w, None, 6, 6), 6, 42, 3)
    counts = np.bincount(acts.cpu().numpy(), minlength=6).astype(np.float64)
    p = torch.softmax(row.to(torch.bfloat16).double(), 0).cpu().numpy()
    chi2 = ((counts - B * p) ** 2 / (B * p)).sum()
    assert chi2 < 30.0, (chi2, counts, B * p)


@pytest.mark.parametrize("A,K1,B", [(170, 512, 2500), (12, 64, 300), (190, 1024, 257), (40, 96, 64), (214, 512, 1000), (222, 128, 70),
                                    (170, 512, 40000), (170, 512, 8192), (97, 256, 33),
                                    # more than one trip of mid_head_sample_kernel's grid (2 workgroups per CU x 4 waves x 32 envs = 65 536
                                    # envs per trip on 256 CUs): exactly one trip, a ragged second one, 7 and 1 action tiles, three trips
                                    (170, 512, 65536), (170, 512, 70001), (214, 512, 66000), (12, 512, 65600), (95, 512, 140000)])
# <= 8 192 envs and in_features % 128 == 0: mid_head_small_kernel
def test_mid_head_sample_on_integer_data_is_exact(A, K1, B):
    """qg_policy_mid_head_sample = relu(h W2^T + b2) -> head -> draw, everything in registers.  Small-integer weights keep
    every intermediate exactly representable (h2 <= 256 in bf16), which pins the fragment k orders: the draw must be
    the race winner on the exact logits, log-prob / entropy / value to f32 round-off."""
    from qiskit_gym_amd.collector import mid_head_sample, pack_head, pack_mid

    seed, counter, F = 5, 11, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(A + K1)
    h1 = torch.randint(0, 3, (B, K1), device="cuda", generator=g).to(torch.bfloat16)
    w2 = torch.zeros((F, K1), device="cuda")
    for r in range(F):  # <= 24 entries of +-1 per feature: |h2| <= 48 + bias
        idx = torch.randperm(K1, device="cuda", generator=g)[:24]
        w2[r, idx] = (torch.randint(0, 2, (idx.numel(),), device="cuda", generator=g) * 2 - 1).float()
    b2 = torch.randint(-4, 5, (F,), device="cuda", generator=g).float()
    w3 = torch.zeros((A + 2, F), device="cuda")
    for r in range(A + 2):
        idx = torch.randperm(F, device="cuda", generator=g)[:6]
        w3[r, idx] = (torch.randint(0, 2, (idx.numel(),), device="cuda", generator=g) * 2 - 1).float() * 0.125
    b3 = torch.randint(-3, 4, (A + 2,), device="cuda", generator=g).float() * 0.25
    value_row = A + 1
    acts, logp, ent, vals = mid_head_sample(h1, pack_mid(w2, b2), F, pack_head(w3, b3, A, value_row, after_mid=True), A, seed, counter)
    torch.cuda.synchronize()
    h2 = torch.relu(h1.double() @ w2.double().t() + b2.double())
    assert float(h2.max()) <= 256
    full = (h2 @ w3.double().t() + b3.double()).cpu().numpy()
    logits = full[:, :A]
    keys = race_keys(logits, sample_uniforms(seed, B, counter, A))
    order = np.sort(keys, axis=1)
    margin = order[:, 1] - order[:, 0]
    got = acts.cpu().numpy()
    clear = margin > 1e-4
    assert clear.mean() > 0.99
    np.testing.assert_array_equal(got[clear], keys.argmin(axis=1)[clear])
    lsm = log_softmax(logits)
    np.testing.assert_allclose(logp.cpu().numpy(), lsm[np.arange(B), got], rtol=0, atol=3e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), -(np.exp(lsm) * lsm).sum(axis=1), rtol=0, atol=1e-4)
    np.testing.assert_array_equal(vals.cpu().numpy(), full[:, value_row].astype(np.float32))


@pytest.mark.parametrize("A,K1", [(170, 512), (1, 128), (31, 256), (32, 1024), (33, 128), (95, 512), (128, 384), (222, 512), (223 - 32, 640)])
def test_mid_head_sample_small_and_large_batch_kernels_draw_the_same_actions(A, K1):
    """Up to 8 192 envs (one 32-env workgroup per CU) the launch takes mid_head_small_kernel, beyond mid_head_sample_kernel: same
    weights, same k order, same race keys -- identical actions and values for the same env ids; log-prob / entropy sum in another order.
    Every count of action tiles (1..7, the value head in each position of a wave's tiles) and in_features of 1..8 register groups."""
    from qiskit_gym_amd.collector import mid_head_sample, pack_head, pack_mid

    F, B = 256, 8192 + 2048
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    h1 = torch.randn((B, K1), device="cuda", generator=g).clamp_min(0).to(torch.bfloat16)
    w2 = (torch.randn((F, K1), device="cuda", generator=g) * (2.0 / K1) ** 0.5).to(torch.bfloat16)
    b2 = (torch.randn(F, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    w3 = (torch.randn((A + 1, F), device="cuda", generator=g) * (2.0 / F) ** 0.5).to(torch.bfloat16)
    b3 = (torch.randn(A + 1, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    pm, ph = pack_mid(w2, b2), pack_head(w3, b3, A, A, after_mid=True)
    big = mid_head_sample(h1, pm, F, ph, A, 7, 3)
    for n in (8192, 1024, 77) if (A, K1) == (170, 512) else (333,):
        small = mid_head_sample(h1[:n].contiguous(), pm, F, ph, A, 7, 3)
        torch.cuda.synchronize()
        assert torch.equal(small[0], big[0][:n])
        assert torch.equal(small[3], big[3][:n])
        torch.testing.assert_close(small[1], big[1][:n], rtol=0, atol=2e-6)
        torch.testing.assert_close(small[2], big[2][:n], rtol=0, atol=2e-6)


def test_mid_head_sample_random_weights():
    """Random bf16 weights: against an f64 reference that rounds h2 to bf16 like the kernel does."""
    from qiskit_gym_amd.collector import mid_head_sample, pack_head, pack_mid

    A, K1, F, B = 170, 512, 256, 4000
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    h1 = torch.randn((B, K1), device="cuda", generator=g).clamp_min(0).to(torch.bfloat16)
    w2 = (torch.randn((F, K1), device="cuda", generator=g) * (2.0 / K1) ** 0.5).to(torch.bfloat16)
    b2 = (torch.randn(F, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    w3 = (torch.randn((A + 1, F), device="cuda", generator=g) * (2.0 / F) ** 0.5).to(torch.bfloat16)
    b3 = (torch.randn(A + 1, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    acts, logp, ent, vals = mid_head_sample(h1, pack_mid(w2, b2), F, pack_head(w3, b3, A, A, after_mid=True), A, 1, 2)
    torch.cuda.synchronize()
    h2 = torch.relu(h1.double() @ w2.double().t() + b2.double()).to(torch.bfloat16).double()
    full = (h2 @ w3.double().t() + b3.double()).cpu().numpy()
    logits = full[:, :A]
    keys = race_keys(logits, sample_uniforms(1, B, 2, A))
    order = np.sort(keys, axis=1)
    clear = (order[:, 1] - order[:, 0]) > 5e-2  # an h2 element on a bf16 rounding boundary moves a logit by ~1e-3
    got = acts.cpu().numpy()
    assert clear.mean() > 0.85
    np.testing.assert_array_equal(got[clear], keys.argmin(axis=1)[clear])
    lsm = log_softmax(logits)
    np.testing.assert_allclose(logp.cpu().numpy(), lsm[np.arange(B), got], rtol=0, atol=2e-2)
    np.testing.assert_allclose(vals.cpu().numpy(), full[:, A], rtol=0, atol=2e-2)
