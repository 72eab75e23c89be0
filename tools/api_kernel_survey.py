"""Every batched entry point once per loop, for rocprofv3 --kernel-trace --stats: which kernel each call launches and what it costs
(looking for a path whose kernel is out of proportion with the bytes it moves).  KIND=clifford|linear_function|permutation|pauli, N, B from the env."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

kind = os.environ.get("KIND", "clifford"); n = int(os.environ.get("N", "16")); B = int(os.environ.get("B", "65536"))
inverts = os.environ.get("INVERTS", "0") == "1"
gs = line_gateset(kind, n)
cfg = dict(add_perms=False, track_solution=inverts, difficulty=2 * n)
if kind != "pauli":
    cfg["add_inverts"] = inverts
env = VecEnv(kind, n, gs, B, **cfg)
g = torch.Generator(device="cuda").manual_seed(0)
for it in range(6):
    env.reset(it)
    acts = torch.randint(0, len(gs), (B,), dtype=torch.int32, device="cuda", generator=g)
    env.step(acts)
    env.observe()
    env.observe_packed()
    env.masks()
    if kind != "pauli":
        st = env.get_state("packed")
        wire = env.get_state("i64")
        env.set_state(st, "packed")
        env.set_state(wire, "i64")  # the trait's Vec<i64> (clifford.rs:299-304)
    else:
        env.get_state("i64")
    env.reset_done(it + 100)
    env.pack_learner_shard()
    env.sync()
torch.cuda.synchronize()
print("done")
