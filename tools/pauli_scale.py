"""PauliGym 20q step time against the batch size (is the step latency- or throughput-bound?)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset
from test_gpu_pauli import random_labels, random_tableau

n, T = 20, 128
gs = line_gateset("pauli", n); A = len(gs)
pairs = [g[1] for g in gs if g[0] == "CX"]
rng = np.random.default_rng(5); U = 128
tabs = [random_tableau(rng, n, 256, pairs) for _ in range(U)]
labs = [random_labels(rng, n, int(rng.integers(1, 8)), 4) for _ in range(U)]
for kind in ("pauli", "clifford"):
    for B in (16384, 32768, 65536, 131072, 262144, 524288):
        if kind == "pauli":
            env = VecEnv("pauli", n, gs, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=128)
            env.pauli_reset_from(np.stack([tabs[e % U] for e in range(B)]), [labs[e % U] for e in range(B)])
            acts = torch.randint(0, A, (16, B), dtype=torch.int32, device="cuda")
        else:
            g16 = line_gateset("clifford", 16)
            env = VecEnv("clifford", 16, g16, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
            env.reset(1)
            acts = torch.randint(0, len(g16), (16, B), dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            env.rollout_ring(acts, T); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(8): env.rollout_ring(acts, T)
            e1.record(s)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (8 * T)
        print(f"{kind:9s} B={B:7d} ({B / 65536:4.2f} waves/SIMD) {us:7.2f} us/step  {B / us * 1e6:.3e} env-steps/s")
        env.close()
