"""qg_vec_reset_done of CliffordGym 16q x B envs (difficulty 256) at finer fractions of finished envs than tools/bench_reset_done.py: where the
tree, the 16-lane form and the flat form trade places (TREE_MAX_ENVS, qgym_plan.hpp).  Development."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

for kind, n, diff in (("clifford", 16, 256), ("clifford", 24, 256)):
    gs = line_gateset(kind, n)
    B = int(os.environ.get("B", "65536"))
    env = VecEnv(kind, n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    env.reset(1)
    for count in (256, 512, 1024, 1500, 2048, 3000, 4096, 8192):
        if count > B:
            continue
        idx = torch.randperm(B, device="cuda")[:count]
        mask = torch.zeros(B, dtype=torch.uint8, device="cuda"); mask[idx] = 1
        times = []
        for i in range(12):
            env.done.copy_(mask)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.reset_done(100 + i); e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        times.sort()
        print(f"{kind}{n} difficulty {diff}: {count:5d} of {B} envs finished -> reset_done {times[len(times) // 2]:8.1f} us")
    env.sync(); env.close()
