"""qg_vec_reset_done / qg_vec_reset_done_step on the one-word layouts (LinearFunctionGym 8q = config 2's env, PermutationGym 3x3): eager call
times by finished fraction, and the auto-reset pair (reset_done + step) as a captured graph against the plain step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import grid_gateset, line_gateset

T = 128
for kind, n, diff, B in (("linear_function", 8, 64, 65536), ("linear_function", 8, 64, 8192), ("permutation", 9, 16, 65536)):
    gs = grid_gateset("permutation", 3, 3) if kind == "permutation" else line_gateset(kind, n)
    env = VecEnv(kind, n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=diff)
    env.reset(1)
    for frac in (1.0, 0.5, 0.1, 0.03, 0.01, 0.0):
        mask = (torch.rand(B, device="cuda") < frac).to(torch.uint8)
        times = []
        for i in range(12):
            env.done.copy_(mask)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.reset_done(100 + i)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        times.sort()
        print(f"{kind}{n} x {B} difficulty {diff:3d}: {frac * 100:5.1f} % done -> reset_done (eager) {times[len(times) // 2]:8.1f} us", flush=True)
    # the pair in a graph, episode ends spread evenly over time as a collector sees them: class k = {env : env % L == k} is reset at warm-up step k
    # (L = the episode length), so ~1 / L of the batch finishes in every step
    L = min(2 * diff, 128)
    acts = torch.randint(0, len(gs), (T, B), dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    cls = torch.arange(B, device="cuda") % L
    fin = torch.zeros((T, B), dtype=torch.uint8, device="cuda")
    for name, body in (("step only", lambda t: env.rollout(acts[t:t + 1], dones_out=fin[t:t + 1])),
                       ("reset_done_step", lambda t: env.reset_done_step(1000 + t, acts[t], dones_out=fin[t])),
                       ("reset_done + step (two calls)", lambda t: (env.reset_done(1000 + t), env.rollout(acts[t:t + 1], dones_out=fin[t:t + 1])))):
        with torch.cuda.stream(stream):
            env.reset(5)
            for k in range(L):
                env.step(acts[k % T])
                env.reset_done(7000 + k)
                env.done[cls == k] = 1
                env.reset_done(8000 + k)
            for t in range(T):
                body(t)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for t in range(T):
                    body(t)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(4):
                g.replay()
            e1.record(stream)
            torch.cuda.synchronize()
        per = fin.float().mean(dim=1)
        print(f"{kind}{n} x {B}: {name:32s} {e0.elapsed_time(e1) * 1e3 / (4 * T):6.2f} us per step (graph of {T}); finished per step "
              f"{float(per.mean()):.4f} [{float(per.min()):.4f}, {float(per.max()):.4f}]", flush=True)
    env.sync()
    env.close()
