"""Run ONE BASELINE.json configuration's env.step() loop and nothing else (one step kernel, one batch size), so that
a rocprofv3 --kernel-trace --stats / --pmc pass over this command holds that kernel alone:

  python tools/run_config.py --config C3 [--envs 4194304] [--steps 512]

Prints one JSON line with the live launch period (HIP events on the launch stream around hipGraph replays).
Configurations follow SURVEY.md 8(d): C1 PermutationGym 3x3 x 128, C2 LinearFunctionGym 8q x 8 192, C3 CliffordGym 16q x 65 536,
C5 PauliGym 20q x 65 536 (1-7 rotations per env, tableau scrambled by 256 gates); C3d = C3 with the reference's default
add_inverts=True / track_solution=True.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
if os.environ.get('QG_LIB'):  # development: a variant build
    from qiskit_gym_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['QG_LIB'])

from qiskit_gym_amd.vec import VecEnv
from util import grid_gateset, line_gateset

ALGO = {"C1": 32, "C2": 32, "C3": 160, "C3d": 160, "C5": 494, "LFd": 0}  # SURVEY.md 8(d), bytes per env-step
KERNELS = {"C1": "word_step_kernel<true>", "C2": "word_step_kernel<false>", "C3": "qm_step1_kernel<16, true, false, *>",
           "C3d": "qm_inv2_kernel<16, true, *>", "C5": "ptile_step1c_kernel<20, 8, *>", "LFd": "lfd_step_kernel"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=sorted(ALGO))
    ap.add_argument("--envs", type=int, default=None)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--qubits", type=int, default=32, help="LFd: LinearFunctionGym with the reference defaults at this many qubits")
    ap.add_argument("--no-track", action="store_true", help="C3d: add_inverts only (no solution log)")
    ap.add_argument("--coin", default="rand", choices=["rand", "0", "1"], help="C3d: the inversion coins (ablation: never / always invert)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    plain = dict(add_inverts=False, add_perms=False, track_solution=False)
    c = args.config
    coins = None
    if c == "C1":
        gs, B = grid_gateset("permutation", 3, 3), args.envs or 128
        env = VecEnv("permutation", 9, gs, B, difficulty=16, **plain)
        env.reset(0x5EED0001)
    elif c == "C2":
        gs, B = line_gateset("linear_function", 8), args.envs or 8192
        env = VecEnv("linear_function", 8, gs, B, difficulty=64, **plain)
        env.reset(0x5EED0002)
    elif c == "LFd":
        gs, B = line_gateset("linear_function", args.qubits), args.envs or 65536
        env = VecEnv("linear_function", args.qubits, gs, B, difficulty=64, add_inverts=True, add_perms=False, track_solution=not args.no_track, max_depth=args.chunk)
        coins = torch.randint(0, 2, (args.chunk, B), dtype=torch.uint8, device=dev)
        if args.coin != "rand":
            coins.fill_(int(args.coin))
        env.reset(0x5EED0002)
    elif c in ("C3", "C3d"):
        gs, B = line_gateset("clifford", 16), args.envs or 65536
        if c == "C3":
            env = VecEnv("clifford", 16, gs, B, difficulty=256, **plain)
        else:
            env = VecEnv("clifford", 16, gs, B, difficulty=256, add_inverts=True, add_perms=False, track_solution=not args.no_track, max_depth=args.chunk)
            coins = torch.randint(0, 2, (args.chunk, B), dtype=torch.uint8, device=dev)
            if args.coin != "rand":
                coins.fill_(int(args.coin))
        env.reset(0x5EED0003)
    else:  # C5: every env's target generated on the device (1-7 rotations, tableau scrambled by 256 gates; tests/test_gpu_fullsize.py checks it)
        n, B = 20, args.envs or 65536
        gs = line_gateset("pauli", n)
        env = VecEnv("pauli", n, gs, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8)
        env.reset(0x5EED0005)
    A = len(gs)
    T = args.chunk
    stream = torch.cuda.Stream(device=dev)
    acts = torch.randint(0, A, (16 if coins is None else T, B), dtype=torch.int32, device=dev)
    reps = max(1, args.steps // T)
    with torch.cuda.stream(stream):
        def run():
            if coins is None:
                env.rollout_ring(acts, T)
            else:
                env.reset(0x5EED0003)  # a fresh episode: the solution log holds max_depth entries
                env.rollout(acts, coins=coins)
        run()
        torch.cuda.synchronize()
        total = 0.0
        for _ in range(reps):
            if coins is not None:
                env.reset(0x5EED0003)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if coins is None:
                env.rollout_ring(acts, T)
            else:
                env.rollout(acts, coins=coins)
            e1.record(stream)
            torch.cuda.synchronize()
            total += e0.elapsed_time(e1)
    try:
        env.sync()
    except Exception as exc:  # (an ablation build computes wrong states on purpose: its envs may fault)
        if not os.environ.get("QG_LIB"):
            raise
        print(f"run_config: variant build, faults ignored: {exc}", file=sys.stderr)
    us = total * 1e3 / (reps * T)
    print(json.dumps({"config": c, "envs": B, "actions": A, "kernel": "qg::" + KERNELS[c], "launch_us": us, "env_steps_per_s": B / us * 1e6,
                      "algorithmic_bytes_per_env_step": ALGO[c], "achieved_GBs": ALGO[c] * B / us / 1e3, "frac_of_8TBs": ALGO[c] * B / us / 1e3 / 8000,
                      "steps_timed": reps * T, "launch": f"hipGraph replays of {T} single-step launches"}))


if __name__ == "__main__":
    main()
