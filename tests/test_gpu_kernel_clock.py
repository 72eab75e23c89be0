"""qg_vec_set_kernel_clock (include/qgym.h): the kernel's own waves stamp first entry / last exit of every launch on the device clock.
The stamps must not change any result, every stamped launch must report a duration that fits inside the launch period the host sees,
and launches past the last slot (or kernels without stamps) must leave their slots alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from util import line_gateset  # noqa: E402


@pytest.mark.parametrize("kind,n,B,cfg", [
    ("clifford", 16, 65536, dict(add_inverts=False, track_solution=False)),   # qm_step1_kernel (the headline)
    ("clifford", 16, 4100, dict(add_inverts=True, track_solution=True)),      # qm_inv2_kernel, ragged last wave
    ("linear_function", 8, 8192, dict(add_inverts=False, track_solution=False)),  # word_step_kernel
    ("pauli", 20, 5000, dict(track_solution=False, max_rotations=5, pauli_diff_scale=8)),  # ptile_step1c_kernel
    ("clifford", 24, 1000, dict(add_inverts=False, track_solution=False)),    # q64_step1_kernel
    ("linear_function", 16, 3000, dict(add_inverts=True, track_solution=False)),  # lfd_step_kernel
])
def test_stamped_launches_report_durations_and_change_nothing(kind, n, B, cfg):
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset(kind, n)
    A, T = len(gs), 24
    envs = [VecEnv(kind, n, gs, B, add_perms=False, difficulty=32, seed=9, **cfg) for _ in range(2)]
    gen = torch.Generator(device="cuda").manual_seed(2)
    acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda", generator=gen)
    for e in envs:
        e.reset(4)
    slots = envs[0].kernel_clock(T - 4)  # the last four launches have no slot
    untouched = slots.clone()
    t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0[0].record()
    envs[0].rollout(acts, fused=False)
    t0[1].record()
    envs[1].rollout(acts, fused=False)
    for e in envs:
        e.sync()
    d = envs[0].kernel_durations_us(slots)
    assert len(d) == T - 4, "every slotted launch stamps its slot"
    assert (d > 0.3).all() and d.sum() < t0[0].elapsed_time(t0[1]) * 1e3, (d.min(), d.sum(), t0[0].elapsed_time(t0[1]) * 1e3)
    live = slots[: T - 4, :, 1] != 0
    first = torch.where(live, slots[: T - 4, :, 0], torch.full_like(slots[: T - 4, :, 0], 2**62)).amin(dim=1)
    last = slots[: T - 4, :, 1].amax(dim=1)
    assert bool((first[1:] >= last[:-1]).all()), "launches on one stream: each starts after the one before has finished"
    assert int(live[0].sum()) >= (B + 63) // 64, "every wave of the grid writes its record"
    fmt = "i64" if kind == "pauli" else "packed"
    assert torch.equal(envs[0].get_state(fmt), envs[1].get_state(fmt))
    assert torch.equal(envs[0].reward.view(torch.int32), envs[1].reward.view(torch.int32)) and torch.equal(envs[0].depth, envs[1].depth)
    # detached: nothing is written any more
    envs[0].kernel_clock(0)
    before = slots.clone()
    envs[0].step(acts[0])
    envs[0].sync()
    assert torch.equal(slots, before) and not torch.equal(slots, untouched)


def test_dense_rewrite_and_reset_launches_are_stamped_and_other_launches_leave_their_slot_alone():
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset("clifford", 16)
    env = VecEnv("clifford", 16, gs, 8192, add_inverts=False, add_perms=False, track_solution=False, difficulty=16)
    env.reset(1)
    slots = env.kernel_clock(5)
    out = env.observe()          # slot 0: qm_dense_stream_kernel
    env.reset(2)                 # slot 1: qm_init_kernel
    env.observe(out=out)         # slot 2
    env.get_state("i64")         # slot 3, taken by an export kernel without stamps
    env.sync()
    d = env.kernel_durations_us(slots)
    assert len(d) == 3 and (d > 0.3).all()
    assert bool(slots[:3].any(dim=2).any(dim=1).all()) and not bool(slots[3:].any())
