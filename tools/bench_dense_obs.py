"""SURVEY 8(d)'s two observation modes of the headline workload (CliffordGym 16q x 65 536, free-running), each as one hipGraph:
  packed   K x (qg_vec_step, qg_vec_observe_packed)          160 + 128 B/env-step written out
  dense    K x (qg_vec_step, qg_vec_observe_dense)           1 184 B/env-step (8d): a full 1 KiB rewrite per env and step
  tracked  K x qg_vec_step on a handle with qg_vec_track_dense  the step rewrites the <= 4 rows its gate changed (<= 128 B)
Run under rocprofv3 --kernel-trace --stats for the per-kernel durations (tools/profile_dense.sh)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from qiskit_gym_amd.vec import VecEnv  # noqa: E402
from util import line_gateset  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--replays", type=int, default=8)
ap.add_argument("--inverts", action="store_true", help="the reference-default add_inverts=True (tracked mode = step + full rewrite)")
ap.add_argument("--modes", default="step,packed,dense,tracked")
ap.add_argument("--lib", default=None, help="development: a variant build of libqgym.so (tools/build_variant.sh)")
args = ap.parse_args()
if args.lib:
    from qiskit_gym_amd import _lib

    _lib.LIB_PATH = os.path.abspath(args.lib)

gs = line_gateset("clifford", 16)
A, B, K = len(gs), args.envs, args.steps
stream = torch.cuda.Stream()
out = {}
for mode in args.modes.split(","):
    env = VecEnv("clifford", 16, gs, B, add_inverts=args.inverts, add_perms=False, track_solution=False, difficulty=256)
    acts = torch.randint(0, A, (16, B), dtype=torch.int32, device="cuda")
    coins = torch.randint(0, 2, (16, B), dtype=torch.uint8, device="cuda") if args.inverts else None
    obs_d = torch.empty((B, 32, 32), dtype=torch.int8, device="cuda")
    obs_p = torch.empty((B, 32), dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        env.reset(5)
        if mode == "tracked":
            obs_d = env.track_dense()

        def body():
            for t in range(K):
                env.step(acts[t % 16], None if coins is None else coins[t % 16])
                if mode == "packed":
                    env.observe_packed(out=obs_p)
                elif mode == "dense":
                    env.observe(out=obs_d)

        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            body()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.replays):
            g.replay()
        e1.record(stream)
    torch.cuda.synchronize()
    env.sync()
    us = e0.elapsed_time(e1) * 1e3 / (args.replays * K)
    if mode in ("dense", "tracked"):
        assert torch.equal(obs_d, env.observe()), mode
    out[mode] = {"us_per_step": us, "env_steps_per_s": B / (us * 1e-6)}
    del g, env
print(json.dumps({"envs": B, "steps_per_graph": K, "add_inverts": args.inverts, "modes": out}))
