import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset
gs = line_gateset("clifford", 16)
A, B = len(gs), 8192
for T in (16, 7):
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=3, depth_slope=2, max_depth=128)
    g_env, twin = VecEnv("clifford", 16, gs, B, **cfg), VecEnv("clifford", 16, gs, B, **cfg)
    acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        g_env.reset(1); twin.reset(1)
        def body(env):
            for t in range(T):
                env.reset_done_step(40 + t, acts[t])
        body(g_env); body(twin)
        torch.cuda.synchronize()
        print(T, "after eager pass equal:", torch.equal(g_env.get_state("packed"), twin.get_state("packed")))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            body(g_env)
        for r in range(3):
            graph.replay(); body(twin)
            torch.cuda.synchronize()
            sa, sb = g_env.get_state("packed"), twin.get_state("packed")
            bad = (sa != sb).any(dim=1).nonzero().flatten()
            print(T, "replay", r, "mismatching envs:", len(bad), bad[:10].tolist(), "depth diff", int((g_env.depth != twin.depth).sum()), "done diff", int((g_env.done != twin.done).sum()))
