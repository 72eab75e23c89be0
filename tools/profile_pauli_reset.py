"""PauliGym 20q x 65 536: 20 x qg_vec_reset_done with 1 % of the batch finished, for rocprofv3 --kernel-trace --stats (QG_LIB: a variant build).  Development."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if os.environ.get('QG_LIB'):
    from qiskit_gym_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['QG_LIB'])
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset
B = 65536
gs = line_gateset("pauli", 20)
env = VecEnv("pauli", 20, gs, B, add_perms=False, track_solution=False, difficulty=int(os.environ.get("DIFF", "128")))
env.reset(1)
mask = (torch.rand(B, device="cuda") < 0.01).to(torch.uint8)
for i in range(20):
    env.done.copy_(mask); env.reset_done(100 + i)
torch.cuda.synchronize(); env.sync(); print("ok")
