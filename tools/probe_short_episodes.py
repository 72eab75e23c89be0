"""Development: the auto-reset pair when episodes are SHORT (a curriculum's early difficulty: scrambles of a few gates, episodes of 2 x difficulty steps, so a large
share of the batch finishes in every step and the resets take the 16-lane / per-lane paths, not trees): qg_vec_reset_done_step against reset_done + step, one hipGraph
of 64 pairs each, CliffordGym 16q (plain, reference defaults) and 24q x 65 536 envs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

B, T = 65536, 64
import sys as _sys
CASES = ((("linear_function", 8, False), ("linear_function", 8, True), ("permutation", 9, False)) if "--words" in _sys.argv
         else (("clifford", 16, False), ("clifford", 16, True), ("clifford", 24, False)))
for kind, n, inverts in CASES:
    gs = line_gateset(kind, n)
    A = len(gs)
    for diff in (1, 4, 16, 64):
        res = {}
        for fused in (True, False):
            env = VecEnv(kind, n, gs, B, add_inverts=inverts, add_perms=False, track_solution=inverts, difficulty=diff, depth_slope=2, max_depth=128)
            stream = torch.cuda.Stream()
            acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda")
            fin = torch.empty((T, B), dtype=torch.uint8, device="cuda")
            with torch.cuda.stream(stream):
                env.reset(1)
                # spread the episode ends: class k of 2 * diff starts at step k
                L = 2 * diff
                cls = torch.arange(B, device="cuda") % L
                for k in range(L):
                    env.step(acts[k % T])
                    env.reset_done(10 + k)
                    env.done[cls == k] = 1
                    env.reset_done(500 + k)

                def body():
                    for t in range(T):
                        if fused:
                            env.reset_done_step(1000 + t, acts[t], dones_out=fin[t])
                        else:
                            env.reset_done(1000 + t)
                            env.rollout(acts[t:t + 1], dones_out=fin[t:t + 1])
                body()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    body()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(4):
                        g.replay()
                    e1.record(stream)
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * T))
            res[fused] = (best, float(fin.float().mean()))
            env.close()
        print(f"{kind}{n} add_inverts={inverts} difficulty {diff:3d} (episodes of {2 * diff} steps, {res[True][1] * 100:.1f} % finishing per step): one call {res[True][0]:7.2f} us, two calls {res[False][0]:7.2f} us", flush=True)
