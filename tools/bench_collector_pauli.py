"""Collection rate with a policy in the loop for BASELINE config 5's env: PauliGym 20q, line coupling, 65 536 envs, bf16 BasicPolicy
(1800 -> 512 -> 256 -> {214, 1}).  The first layer reads the packed observation words the rollout stores (qg_policy_embed_words; EMBED=0: words
expanded to bf16 + library GEMM); middle layer + head + draw run in qg_policy_mid_head_sample.  GRAPH=1: the collection replayed from one hipGraph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

B = int(os.environ.get("B", "65536"))
gs = line_gateset("pauli", 20)
env = VecEnv("pauli", 20, gs, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=int(os.environ.get("DIFF", "16")))
r, c = env.obs_shape_
embed = {"1": True, "0": False}.get(os.environ.get("EMBED", ""), None)
graph = os.environ.get("GRAPH", "") == "1"
col = RolloutCollector(env, BasicPolicy(r * c, len(gs)), dtype=torch.bfloat16, seed=1, store_obs=os.environ.get("STORE", "packed"), use_bit_embedding=embed,
                       use_graph=graph)
T = 16
ro = col.collect(T)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ro = col.collect(T) if graph else col.collect(T, out=ro)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
env.sync()
print(f"PauliGym 20q B={B}{' hipGraph' if graph else ''}: {3 * T * B / dt:.3e} env-steps/s with the policy in the loop ({dt / (3 * T) * 1e6:.0f} us per step), {float(ro.dones.float().mean()):.3f} done/step")
