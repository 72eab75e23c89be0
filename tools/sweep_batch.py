"""The headline step kernel (CliffordGym 16q, one launch per env.step(), hipGraph of 128 launches over 16 resident action buffers) and the
fused rollout across batch sizes: where the launch boundary stops mattering.  Run on the GPU box: python tools/sweep_batch.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from measure_configs import gpu_rate  # noqa: E402
from qiskit_gym_amd.vec import VecEnv  # noqa: E402
from util import line_gateset  # noqa: E402

gs = line_gateset("clifford", 16)
print("| envs | us / step | env-steps/s | algorithmic GB/s (160 B / env-step) | frac of 8 TB/s | fused rollout env-steps/s |")
print("|---|---|---|---|---|---|")
for lg in range(10, 25, 2):
    B = 1 << lg
    env = VecEnv("clifford", 16, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
    env.reset(1)
    us, rate = gpu_rate(env, len(gs))
    _, frate = gpu_rate(env, len(gs), fused=True, reps=2 if lg >= 22 else 8)
    env.sync()
    print(f"| 2^{lg} = {B} | {us:.2f} | {rate:.3e} | {160 * rate / 1e9:.0f} | {160 * rate / 8e12:.3f} | {frate:.3e} |", flush=True)
    del env
    torch.cuda.empty_cache()
