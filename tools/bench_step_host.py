"""PCIe-inclusive rate of the host-pointer boundary: qg_vec_step_host (actions in, rewards / is_final / success out, pinned host buffers) on the
headline workload (CliffordGym 16q x 65 536 envs), beside qg_vec_step on resident buffers.  Run on the GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiskit_gym_amd import _lib
from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map, line_edges
from qiskit_gym_amd.vec import VecEnv

n, B, T = 16, 65536, 512
gs = gateset_from_coupling_map(line_edges(n, True), None, ["H", "S", "Sdg", "SX", "SXdg", "CX", "CZ", "SWAP"])[1]
env = VecEnv("clifford", n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256, seed=1)
env.reset(1)
L = env._L
g = torch.Generator().manual_seed(0)
acts_h = torch.randint(0, len(gs), (B,), dtype=torch.int32, generator=g).pin_memory()
rew_h = torch.empty(B, dtype=torch.float32).pin_memory()
done_h = torch.empty(B, dtype=torch.uint8).pin_memory()
suc_h = torch.empty(B, dtype=torch.uint8).pin_memory()
acts_d = acts_h.cuda()
s = env._stream()


def host_step(outs=True):
    _lib.check(L.qg_vec_step_host(env._h, acts_h.data_ptr(), 0, None, rew_h.data_ptr() if outs else None, done_h.data_ptr() if outs else None,
                                  suc_h.data_ptr() if outs else None, s))


for name, fn, sync_each in (("qg_vec_step, resident buffers, free-running", lambda: env.step(acts_d), False),
                            ("qg_vec_step_host, actions in + reward / is_final / success out, free-running", host_step, False),
                            ("qg_vec_step_host, actions in only, free-running", lambda: host_step(False), False),
                            ("qg_vec_step_host, all buffers, qg_vec_sync after every step (a host agent that reads the results)", host_step, True)):
    for _ in range(32):
        fn()
    env.sync()
    t0 = time.perf_counter()
    for _ in range(T):
        fn()
        if sync_each:
            env.sync()
    env.sync()
    dt = (time.perf_counter() - t0) / T
    bytes_step = B * 4 + (B * 6 if "out" in name or "all" in name else 0)
    print(f"{name}: {dt * 1e6:.1f} us per step = {B / dt:.3e} env-steps/s" + (f" ({bytes_step / 1e6:.2f} MB over PCIe per step)" if "host" in name else ""), flush=True)
