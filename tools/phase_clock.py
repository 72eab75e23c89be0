"""Development: where inside a reset's tree kernel the time goes -- a libqgym built with -DQG_PHASE_CLOCK lets every workgroup stamp the device clock at marked
lines into the kernel-clock slot of its launch (device_common.hpp phase_stamp); this tool makes ONE qg_vec_reset_done over 512 of 65 536 envs per sample and
prints, per phase, how long the workgroups spent in it (percentiles over the workgroups with work), and the breakdown of the slowest workgroups.

  tools/build_variant.sh ... -DQG_PHASE_CLOCK for kernels_qm64.hip and kernels_pauli_tile.hip, linked into one library
  QG_LIB=<that library> python tools/phase_clock.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
if os.environ.get("QG_LIB"):
    from qiskit_gym_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["QG_LIB"])
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

B, PH, NB = 65536, 8, 512
CASES = [("pauli", 20, dict(add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8),
          ["count known", "env + draws in LDS (wave 0)", "env + draws in LDS (wave 4)", "labels done (wave 4)", "scramble done (wave 0)", "rows arrived (wave 4)", "stored (wave 4)"]),
         ("clifford", 24, dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=256),
          ["count known", "env known", "scramble done", "stored"]),
         ("clifford", 16, dict(add_inverts=True, add_perms=False, track_solution=True, difficulty=256),
          ["count known", "env known", "chain done (wave 0)", "products done", "stored"])]
q = lambda x: " ".join(f"{v:6.2f}" for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))
for kind, n, kw, names in CASES:
    gs = line_gateset(kind, n)
    env = VecEnv(kind, n, gs, B, **kw)
    env.reset(1)
    gen = torch.Generator(device="cuda").manual_seed(1)
    mask = torch.zeros(B, dtype=torch.uint8, device="cuda")
    mask[torch.randperm(B, device="cuda", generator=gen)[:NB]] = 1
    for i in range(8):
        env.done.copy_(mask)
        torch.cuda.synchronize()
        view = env.kernel_clock(2)
        env.reset_done(100 + i)
        torch.cuda.synchronize()
        rec = view[0].cpu().numpy().astype(np.int64)  # the tree launch (TILE: the reset's only launch): [waves, 2]
        env.kernel_clock(0)
    W = rec.shape[0]
    body = rec[:W // 2]
    live = body[:, 1] != 0
    t_first, t_last = body[live, 0].min(), body[live, 1].max()
    top = rec[W // 2:, 0][::-1]  # index 8 b + idx
    st = top[:8 * NB].reshape(NB, 8)[:, :len(names)].astype(np.float64)
    st[st == 0] = np.nan
    st = (st - t_first) / 100.0
    print(f"{kind} {n}q x {B}, {NB} finished (list path), last sample; the launch's last wave exits {(t_last - t_first) / 100.0:.2f} us after its first enters; {int(live.sum())} waves")
    print("    stamp (us after the first wave entry), over the workgroups:        min    p10    p50    p90    p99    max")
    for j, name in enumerate(names):
        print(f"    {name:45s} {q(st[:, j][~np.isnan(st[:, j])])}")
    print("    time between consecutive stamps of the same wave / phase:")
    pairs = [(0, 1), (2, 3), (1, 4), (3, 5), (5, 6)] if kind == "pauli" else [(j, j + 1) for j in range(len(names) - 1)]
    for a, b in pairs:
        d = st[:, b] - st[:, a]
        print(f"    {names[a]:30s} -> {names[b]:30s} {q(d[~np.isnan(d)])}")
    last = np.argsort(-np.nan_to_num(st[:, len(names) - 1]))[:5]
    for b in last:
        print(f"    a slowest workgroup ({b}): " + " ".join(f"{v:6.2f}" for v in st[b]))
    env.close()
