import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset
from test_gpu_pauli import random_labels, random_tableau

n, B, T = 20, 65536, 128
gs = line_gateset("pauli", n); A = len(gs)
pairs = [g[1] for g in gs if g[0] == "CX"]
rng = np.random.default_rng(5); U = 128
tabs = [random_tableau(rng, n, 256, pairs) for _ in range(U)]
labs = [random_labels(rng, n, int(rng.integers(1, 8)), 4) for _ in range(U)]
env = VecEnv("pauli", n, gs, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=128)
def reset(): env.pauli_reset_from(np.stack([tabs[e % U] for e in range(B)]), [labs[e % U] for e in range(B)])
def run(name, acts):
    reset()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        env.rollout_ring(acts, T); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(8): env.rollout_ring(acts, T)
        e1.record(s)
    torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) * 1e3 / (8 * T):7.2f} us/step")
kinds = {k: [i for i, g in enumerate(gs) if g[0] == k] for k in ("H", "S", "Sdg", "SX", "CX", "CZ", "SWAP")}
def pick(ids): return torch.as_tensor(np.random.default_rng(1).choice(ids, size=(16, B)), dtype=torch.int32, device="cuda")
run("all actions", torch.randint(0, A, (16, B), dtype=torch.int32, device="cuda"))
run("out-of-range (loads + check only)", torch.full((16, B), A, dtype=torch.int32, device="cuda"))
for k, ids in kinds.items(): run(f"only {k}", pick(ids))
