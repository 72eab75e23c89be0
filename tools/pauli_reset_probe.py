import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset
gs = line_gateset("pauli", 20); B = 65536
for diff, scale in ((8, 8), (32, 8), (128, 8), (256, 8), (128, 1000)):
    env = VecEnv("pauli", 20, gs, B, add_perms=False, track_solution=False, difficulty=diff, pauli_diff_scale=scale)
    env.reset(1)
    for frac in (0.01, 0.0):
        mask = (torch.rand(B, device="cuda") < frac).to(torch.uint8)
        ts = []
        for i in range(8):
            env.done.copy_(mask)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.reset_done(100 + i); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"pauli20 difficulty {diff:3d} pauli_diff_scale {scale}: {frac*100:4.1f} % done -> {ts[len(ts)//2]:7.1f} us")
    env.sync(); env.close()
