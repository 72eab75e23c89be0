"""Development: how long ptile_reset_tree_kernel's phases take (the kernel device clock of the tree launch of ONE qg_vec_reset_done over 512 of 65 536
envs), with the truncated builds of tools/build_variant.sh (-DQG_PT_STOP=1: after the mask / list and the pre-drawn words, 2: after the labels,
3: after the tableau scramble; none: the whole kernel).  The tool raises the flags itself before every call, so a truncated build sees the same work.

  QG_LIB=qiskit_gym_amd/lib/variants/libqgym_ptstop2.so python tools/pauli_tree_phases.py
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

B, n = 65536, 20
gs = line_gateset("pauli", n)
for kw in (dict(max_rotations=5, difficulty=256, pauli_diff_scale=8), dict(max_rotations=5, difficulty=64, pauli_diff_scale=8)):
    env = VecEnv("pauli", n, gs, B, add_perms=False, track_solution=False, **kw)
    env.reset(1)
    gen = torch.Generator(device="cuda").manual_seed(1)
    mask = torch.zeros(B, dtype=torch.uint8, device="cuda")
    mask[torch.randperm(B, device="cuda", generator=gen)[:512]] = 1
    durs = []
    for i in range(24):
        env.done.copy_(mask)
        torch.cuda.synchronize()
        view = env.kernel_clock(4)
        env.reset_done(100 + i)
        torch.cuda.synchronize()
        t0, t1 = view[..., 0], view[..., 1]
        live = t1 != 0
        per = []
        for s in range(4):
            if bool(live[s].any()):
                per.append(float((t1[s].max() - t0[s][live[s]].min()).item()) / 100.0)
        durs.append(per)
        env.kernel_clock(0)
    k = max(len(p) for p in durs)
    med = [float(np.median([p[j] for p in durs[4:] if len(p) > j])) for j in range(k)]
    print(f"pauli20 x {B}, {kw}: 512 finished -> stamped launches of reset_done, us (median of 20): " + ", ".join(f"{m:.2f}" for m in med), flush=True)
    env.close()
