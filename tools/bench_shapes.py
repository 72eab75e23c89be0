"""Development timing of shapes that use the non-headline kernel families."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset
from bench_variants import timeit

B = 65536
base = dict(add_perms=False, difficulty=64, track_solution=False)
for n in (8, 12, 16, 20, 24, 32):
    gs = line_gateset("clifford", n)
    timeit(f"clifford{n} no-inv", VecEnv("clifford", n, gs, B, add_inverts=False, **base), len(gs))
for n in (20, 32):
    gs = line_gateset("clifford", n)
    timeit(f"clifford{n} inverts", VecEnv("clifford", n, gs, B, add_inverts=True, **base), len(gs), coins=True)
for n in (12, 16, 32, 48, 64):
    gs = line_gateset("linear_function", n)
    timeit(f"lf{n} no-inv", VecEnv("linear_function", n, gs, B, add_inverts=False, **base), len(gs))
for n in (12, 32, 64):
    gs = line_gateset("linear_function", n)
    timeit(f"lf{n} inverts", VecEnv("linear_function", n, gs, B, add_inverts=True, **base), len(gs), coins=True)
