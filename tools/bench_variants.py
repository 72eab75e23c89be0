"""Development timing of non-headline configurations (not the judged bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset, grid_gateset


def timeit(name, env, A, T=128, fused=False, reps=4, coins=False):
    B = env.batch
    acts = torch.randint(0, A, (T, B), dtype=torch.int32, device="cuda")
    c = torch.randint(0, 2, (T, B), dtype=torch.uint8, device="cuda") if coins else None
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        env.reset(1)
        env.rollout(acts, fused=fused, coins=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            env.rollout(acts, fused=fused, coins=c)
        e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * T)
    print(f"{name:58s} {us:8.2f} us/step  {B / us * 1e6:.3e} env-steps/s", flush=True)
    try:
        env.sync()
    except Exception as ex:
        print("   (faults:", str(ex)[:80], ")")


if __name__ == "__main__":
    B = 65536
    gs = line_gateset("clifford", 16)
    base = dict(add_perms=False, difficulty=64)
    timeit("clifford16 tile no-inv no-track", VecEnv("clifford", 16, gs, B, add_inverts=False, track_solution=False, **base), len(gs))
    timeit("clifford16 tile no-inv track", VecEnv("clifford", 16, gs, B, add_inverts=False, track_solution=True, **base), len(gs))
    timeit("clifford16 inverts(coins) no-track", VecEnv("clifford", 16, gs, B, add_inverts=True, track_solution=False, **base), len(gs), coins=True)
    timeit("clifford16 inverts(rng) track [reference defaults]", VecEnv("clifford", 16, gs, B, add_inverts=True, track_solution=True, **base), len(gs))
    timeit("clifford16 layers-weighted", VecEnv("clifford", 16, gs, B, add_inverts=False, track_solution=False, metrics_weights={"n_layers": 0.1}, **base), len(gs))
    timeit("clifford16 fused no-inv", VecEnv("clifford", 16, gs, B, add_inverts=False, track_solution=False, **base), len(gs), fused=True)
    g8 = line_gateset("linear_function", 8)
    timeit("lf8 x8192 no-inv", VecEnv("linear_function", 8, g8, 8192, add_inverts=False, track_solution=False, **base), len(g8))
    timeit("lf8 x65536 no-inv", VecEnv("linear_function", 8, g8, B, add_inverts=False, track_solution=False, **base), len(g8))
    timeit("lf8 x65536 inverts", VecEnv("linear_function", 8, g8, B, add_inverts=True, track_solution=False, **base), len(g8), coins=True)
    gp = line_gateset("pauli", 20)
    pe = VecEnv("pauli", 20, gp, B, add_perms=False, track_solution=False)
    # identity targets with rotations would need host generation; time the bare tableau path
    acts = torch.randint(0, len(gp), (128, B), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pe.rollout(acts, fused=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(4):
            pe.rollout(acts, fused=False)
        e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (4 * 128)
    print(f"{'pauli20 x65536 (no rotations)':58s} {us:8.2f} us/step  {B / us * 1e6:.3e} env-steps/s")
