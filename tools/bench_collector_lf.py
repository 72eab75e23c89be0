"""Collection rate with a policy in the loop for LinearFunctionGym with the reference's defaults (add_inverts=True: the LFD layout).
B, QUBITS, GRAPH as in bench_collector_pauli.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

B, NQ = int(os.environ.get("B", "65536")), int(os.environ.get("QUBITS", "16"))
inv = os.environ.get("INVERTS", "1") == "1"
gs = line_gateset("linear_function", NQ)
env = VecEnv("linear_function", NQ, gs, B, add_inverts=inv, add_perms=False, track_solution=False, difficulty=int(os.environ.get("DIFF", "32")))
r, c = env.obs_shape_
graph = os.environ.get("GRAPH", "") == "1"
col = RolloutCollector(env, BasicPolicy(r * c, len(gs)), dtype=torch.bfloat16, seed=1, store_obs="packed", use_graph=graph)
T = 16
ro = col.collect(T)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ro = col.collect(T) if graph else col.collect(T, out=ro)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
env.sync()
kinds = f"embed={'bits' if col._embed is not None else 'words' if col._embed_words is not None else 'library'}, tail={'fused' if col._mid is not None else 'library'}, step={'fused' if col._fused_step else 'own launch'}"
print(f"LinearFunctionGym {NQ}q add_inverts={inv} B={B}{' hipGraph' if graph else ''}: {3 * T * B / dt:.3e} env-steps/s with the policy in the loop ({dt / (3 * T) * 1e6:.0f} us per step) [{kinds}]")
