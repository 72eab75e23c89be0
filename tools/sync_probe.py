"""What the driver's `--steps 20 --warmup 5` timed region costs beyond its 20 kernels (development probe)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset
gs = line_gateset("clifford", 16); B = 65536
env = VecEnv("clifford", 16, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
stream = torch.cuda.Stream()
acts = torch.randint(0, len(gs), (16, B), dtype=torch.int32, device="cuda")
K = 20
with torch.cuda.stream(stream):
    env.reset(1); env.rollout_ring(acts, K); env.rollout_ring(acts, 5)
torch.cuda.synchronize()
def run(mode, reps=12):
    ts = []
    for rep in range(reps):
        with torch.cuda.stream(stream):
            env.reset(1)
            if "eagerwarm" in mode:
                for i in range(5): env.step(acts[i])
            else:
                env.rollout_ring(acts, 5)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            ev0.record(stream)
            if mode.startswith("head"):
                h = int(mode[4:])
                for i in range(h): env.step(acts[i % 16])
                env.rollout_ring(acts, K - h)
            elif "eagersteps" in mode:
                for i in range(K): env.step(acts[i % 16])
            else:
                env.rollout_ring(acts, K)
            ev1.record(stream)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{mode:22s} first {ts[0]*1e6:7.1f} us  then " + " ".join(f"{t*1e6:6.1f}" for t in ts[1:6]) + f"  min {min(ts)*1e6:6.1f}")
with torch.cuda.stream(stream):
    for h in (1, 2, 4):
        env.rollout_ring(acts, K - h)
torch.cuda.synchronize()
for m in ("graph", "head1", "head2", "head4", "graph", "head1", "head2"):
    run(m)
