"""Cost of qg_vec_reset_done when a few percent of the envs, scattered over the batch, are finished."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import grid_gateset, line_gateset

for kind, n, diff in (("clifford", 16, 256), ("clifford", 16, 32), ("pauli", 20, 128), ("clifford", 32, 256), ("linear_function", 8, 64), ("permutation", 9, 16)):
    gs = grid_gateset("permutation", 3, 3) if kind == "permutation" else line_gateset(kind, n)
    B = int(os.environ.get("B", "65536"))
    kw = dict(add_perms=False, track_solution=False, difficulty=diff)
    if kind != "pauli":
        kw["add_inverts"] = False
    env = VecEnv(kind, n, gs, B, **kw)
    env.reset(1)
    for frac in (1.0, 0.5, 0.1, 0.03, 0.01, 0.0):
        mask = (torch.rand(B, device="cuda") < frac).to(torch.uint8)
        times = []
        for i in range(12):
            env.done.copy_(mask)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.reset_done(100 + i)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        times.sort()
        print(f"{kind}{n} difficulty {diff:3d}: {frac * 100:5.1f} % of {B} envs done -> reset_done {times[len(times) // 2]:8.1f} us")
    env.sync()
    env.close()
