"""qg_vec_reset of the whole batch (CliffordGym 16q / 24q x 65 536 envs, 256 scramble gates per env): HIP events around eager calls, median of 12.  Development."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

for kind, n, kw in (("clifford", 16, dict(add_inverts=False)), ("clifford", 16, dict(add_inverts=True)), ("clifford", 24, dict(add_inverts=False)), ("linear_function", 8, dict(add_inverts=False)),
                    ("pauli", 20, dict(max_rotations=5, pauli_diff_scale=8))):
    gs = line_gateset(kind, n)
    B = 65536
    env = VecEnv(kind, n, gs, B, add_perms=False, track_solution=False, difficulty=256 if n > 8 else 64, **kw)
    ts = []
    for i in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.reset(100 + i); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    print(f"{kind}{n} {kw}: reset of {B} envs {ts[len(ts) // 2]:8.1f} us")
    env.close()
