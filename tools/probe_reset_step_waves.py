"""Diagnostic: per-wave entry / exit stamps of qm_reset_step_kernel launches (one launch per auto-reset pair), relative to the launch's first wave entry,
split by the kind of workgroup, plus the device-side gap between consecutive launches (first entry of launch k+1 - last exit of launch k)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

AT, B = 128, 65536
gs = line_gateset("clifford", 16)
A = len(gs)
INVERTS = "--inverts" in sys.argv
for spread in (True, False):
    env = VecEnv("clifford", 16, gs, B, add_inverts=INVERTS, add_perms=False, track_solution=INVERTS, difficulty=256)
    stream = torch.cuda.Stream()
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.randint(0, A, (AT, B), dtype=torch.int32, device="cuda", generator=gen)
    seed = 5
    with torch.cuda.stream(stream):
        env.reset(seed)
        if spread:
            cls = torch.arange(B, device="cuda") % AT
            for k in range(AT):
                env.step(acts[k]); env.reset_done(seed + 77 * (k + 1)); env.done[cls == k] = 1; env.reset_done(seed + 99 * (k + 1))

        def episode():
            env.rollout(acts[0:1])
            for t in range(1, AT):
                env.reset_done_step(seed + 13 * t, acts[t])
            env.reset_done(seed + 13 * AT)

        n_slots = 1 + AT + 2
        episode(); torch.cuda.synchronize()
        slots = env.kernel_clock(n_slots)  # (armed here: the captured launches take slots 0, 1, ...)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            episode()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        slots.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); g.replay(); e1.record(stream)
        torch.cuda.synchronize()
    h = slots.cpu().numpy()
    t0, t1 = h[..., 0], h[..., 1]
    live = t1 != 0
    used = np.nonzero(live.any(axis=1))[0]
    print(f"spread={spread}: graph replay {e0.elapsed_time(e1) * 1e3 / AT:.2f} us per pair; stamped launches {len(used)}")
    first = np.array([t0[k][live[k]].min() for k in used]); last = np.array([t1[k][live[k]].max() for k in used])
    dur = (last - first) / 100.0
    gap = (first[1:] - last[:-1]) / 100.0
    print(f"  kernel duration (device clock): mean {dur[3:-2].mean():.2f} us; gap between launches: mean {gap[3:-2].mean():.2f} min {gap[3:-2].min():.2f} max {gap[3:-2].max():.2f} us; "
          f"period from stamps {(first[-3] - first[3]) / 100.0 / (len(first) - 6):.2f} us")
    k = used[len(used) // 2]
    w = np.nonzero(live[k])[0]
    rel0, rel1 = (t0[k][w] - t0[k][w].min()) / 100.0, (t1[k][w] - t0[k][w].min()) / 100.0
    print(f"  launch {k}: {len(w)} waves stamped, wave ids {w.min()}..{w.max()}")
    nfin = int(round(float(B) / AT)) if spread else 0
    first_reset, step_blocks = 512, (512 if INVERTS else 256)  # the grid: [reset 0..511][step][reset 512..1023]
    wg = w // 4
    is_step = (wg >= first_reset) & (wg < first_reset + step_blocks)
    vblock = np.where(wg < first_reset, wg, wg - step_blocks)
    for name, sel in (("step workgroups", is_step), (f"reset workgroups with work (vblock < {nfin})", ~is_step & (vblock < nfin)),
                      ("reset workgroups without work", ~is_step & (vblock >= nfin))):
        if sel.any():
            print(f"    {name}: {int(sel.sum())} waves; entry min/median/max {rel0[sel].min():.2f} {np.median(rel0[sel]):.2f} {rel0[sel].max():.2f}; "
                  f"exit min/median/max {rel1[sel].min():.2f} {np.median(rel1[sel]):.2f} {rel1[sel].max():.2f}")
    long_ = (rel1 - rel0) > 1.0
    print(f"    waves alive > 1 us: {int(long_.sum())}; their entry median {np.median(rel0[long_]):.2f}, exit median {np.median(rel1[long_]):.2f}, exit max {rel1[long_].max():.2f}")
    env.kernel_clock(0); env.sync(); env.close()
