"""Direct-write hand-over on one rank (world 1): step + push + wait + release per iteration, eager, 65 536 envs (development timing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.distributed import Communicator
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

gs = line_gateset("clifford", 16); B = 65536
env = VecEnv("clifford", 16, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=64)
env.reset(1)
acts = torch.randint(0, len(gs), (B,), dtype=torch.int32, device="cuda")
comm = Communicator(0, 1, device=0, local=True)
h = comm.p2p_export(int(env.shard_layout().bytes)); comm.p2p_open([h])
def it(step):
    if step: env.step(acts)
    comm.push(env); v = comm.wait(); comm.release()
for name, step in (("push+wait+release", False), ("step+push+wait+release", True), ("push+wait+release", False), ("step+push+wait+release", True)):
    for _ in range(4): it(step)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(64): it(step)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 64 * 1e6:.1f} us per iteration")
comm.check()
