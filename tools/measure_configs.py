"""Measure every BASELINE.json configuration: HIP path (graph rollout and fused rollout) next to
the CPU oracle (best OpenMP thread count), with the algorithmic bytes of SURVEY.md section 8(d).
Writes a markdown table; run on the GPU box:  python tools/measure_configs.py > gpurun_out/configs.md
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle import OracleEnv, OracleVec
from qiskit_gym_amd.vec import VecEnv
from test_gpu_pauli import random_labels, random_tableau
from util import grid_gateset, line_gateset


def gpu_rate(env, A, T=128, fused=False, ring=16, reps=8):
    B = env.batch
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        if fused:
            acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda")
            run = lambda: env.rollout(acts, fused=True)
        else:
            acts = torch.randint(0, A, (ring, B), dtype=torch.int32, device="cuda")
            run = lambda: env.rollout_ring(acts, T)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            run()
        e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * T)
    return us, B / us * 1e6


def obs_times(env, reps=50):
    """us per launch of the observation kernels: dense int8 (the Gym format), bf16 (policy input), bit-packed."""
    out = {}
    r, c = env.obs_shape_
    bufs = {"int8": env.observe(), "bf16": env.observe_as(torch.bfloat16)}
    calls = {"int8": lambda: env.observe(out=bufs["int8"]), "bf16": lambda: env.observe_as(torch.bfloat16, out=bufs["bf16"])}
    if env.env_kind != "pauli" or c <= 64:
        bufs["packed"] = env.observe_packed()
        calls["packed"] = lambda: env.observe_packed(out=bufs["packed"])
    for name, fn in calls.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = (e0.elapsed_time(e1) * 1e3 / reps, bufs[name].numel() * bufs[name].element_size())
    return out


OBS_ROWS = []


def cpu_rate(ov, A, budget=4.0):
    B = ov.batch
    rng = np.random.default_rng(0)
    acts = rng.integers(0, A, size=(16, B)).astype(np.int32)
    avail = len(os.sched_getaffinity(0))
    for t in range(4):
        ov.step_only(acts[t], threads=min(8, avail))
    best = (0.0, 1)
    for c in [c for c in (1, 2, 4, 8, 16, 32, 64, 128) if c <= avail]:
        ov.step_only(acts[0], threads=c)
        t0 = time.perf_counter()
        for t in range(3):
            ov.step_only(acts[t], threads=c)
        r = 3 * B / (time.perf_counter() - t0)
        if r > best[0]:
            best = (r, c)
    n = int(max(4, min(2000, budget * best[0] / B)))
    t0 = time.perf_counter()
    for t in range(n):
        ov.step_only(acts[t % 16], threads=best[1])
    return B * n / (time.perf_counter() - t0), best[1]


def main():
    rows = []
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False)
    ocfg = {k: int(v) for k, v in cfg.items()}

    def generic(name, kind, n, gs, B, scramble, algo_bytes, cpu_B=None):
        A = len(gs)
        env = VecEnv(kind, n, gs, B, difficulty=scramble, **cfg)
        env.reset(1)
        us, rate = gpu_rate(env, A)
        OBS_ROWS.append((name, B, us, algo_bytes, obs_times(env)))
        env.reset(1)
        fus, frate = gpu_rate(env, A, fused=True)
        env.sync()
        cb = cpu_B or min(B, 16384)
        proto = OracleEnv(kind, n, gs, difficulty=scramble, **ocfg)
        ov = OracleVec(proto, cb)
        ov.reset_with(np.random.default_rng(1).integers(0, A, size=(scramble, cb)))
        crate, cthreads = cpu_rate(ov, A)
        rows.append((name, B, A, us, rate, algo_bytes, algo_bytes * rate / 1e9, frate, crate, cthreads))
        env.close()

    generic("C1 PermutationGym 3x3 grid", "permutation", 9, grid_gateset("permutation", 3, 3), 128, 16, 8 + 8 + 16, cpu_B=128)
    generic("C2 LinearFunctionGym 8q line", "linear_function", 8, line_gateset("linear_function", 8), 8192, 64, 32, cpu_B=8192)
    generic("C3 CliffordGym 16q line", "clifford", 16, line_gateset("clifford", 16), 65536, 256, 160)
    generic("(C3 shape at 2^20 envs)", "clifford", 16, line_gateset("clifford", 16), 1 << 20, 256, 160)

    # C5 PauliGym 20q
    n, B = 20, 65536
    gs = line_gateset("pauli", n)
    A = len(gs)
    pairs = [g[1] for g in gs if g[0] == "CX"]
    rng = np.random.default_rng(5)
    U = 256
    tabs = [random_tableau(rng, n, 256, pairs) for _ in range(U)]
    labs = [random_labels(rng, n, int(rng.integers(1, 8)), 4) for _ in range(U)]
    pcfg = dict(add_perms=False, track_solution=False, max_rotations=5, difficulty=128)
    env = VecEnv("pauli", n, gs, B, **pcfg)
    env.pauli_reset_from(np.stack([tabs[e % U] for e in range(B)]), [labs[e % U] for e in range(B)])
    us, rate = gpu_rate(env, A)
    OBS_ROWS.append(("C5 PauliGym 20q line, 1-7 rotations", B, us, 494, obs_times(env)))
    env.pauli_reset_from(np.stack([tabs[e % U] for e in range(B)]), [labs[e % U] for e in range(B)])
    fus, frate = gpu_rate(env, A, fused=True)
    env.sync()
    proto = OracleEnv("pauli", n, gs, **{k: int(v) for k, v in pcfg.items()})
    proto.pauli_reset_from(tabs[0], labs[0])
    ov = OracleVec(proto, 4096)
    crate, cthreads = cpu_rate(ov, A)
    rows.append(("C5 PauliGym 20q line, 1-7 rotations", B, A, us, rate, 494, 494 * rate / 1e9, frate, crate, cthreads))

    print("| config | envs | actions | us / step (one launch per step) | env-steps/s | algorithmic B / env-step | achieved GB/s | frac of 8 TB/s | fused rollout env-steps/s | CPU oracle env-steps/s (threads) | GPU / CPU |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, B, A, us, rate, ab, gbs, frate, crate, cthreads in rows:
        print(f"| {name} | {B} | {A} | {us:.2f} | {rate:.3e} | {ab} | {gbs:.0f} | {gbs / 8000:.3f} | {frate:.3e} | {crate:.3e} ({cthreads}) | {rate / crate:.0f}x |")

    print()
    print("Observation kernels (eager launches, event-timed) and SURVEY section 8(d)'s dense-observation mode (one step + one dense int8 observation):")
    print()
    print("| config | envs | observe int8: us, GB/s written | observe bf16: us, GB/s | observe packed: us | step + int8 observe: us | B / env-step | GB/s | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, B, us, ab, ot in OBS_ROWS:
        i8, bf = ot["int8"], ot["bf16"]
        pk = f"{ot['packed'][0]:.1f}" if "packed" in ot else "-"
        per_env = ab + i8[1] // B
        tot = us + i8[0]
        gbs = per_env * B / tot / 1e3
        print(f"| {name} | {B} | {i8[0]:.1f}, {i8[1] / i8[0] / 1e3:.0f} | {bf[0]:.1f}, {bf[1] / bf[0] / 1e3:.0f} | {pk} | {tot:.1f} | {per_env} | {gbs:.0f} | {gbs / 8000:.3f} |")


if __name__ == "__main__":
    main()
