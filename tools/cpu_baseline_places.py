"""bench.py's cpu_baseline leg under different OpenMP placements (development): OMP_PLACES / OMP_PROC_BIND from the environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import OracleEnv, OracleVec
from util import line_gateset
gs = line_gateset("clifford", 16); B = 65536; A = len(gs)
ov = OracleVec(OracleEnv("clifford", 16, gs, add_inverts=0, add_perms=0, track_solution=0, difficulty=256), B)
rng = np.random.default_rng(0); ov.reset_with(rng.integers(0, A, size=(256, B)))
acts = rng.integers(0, A, size=(32, B)).astype(np.int32)
for th in (1, int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    for t in range(8): ov.step_only(acts[t], threads=th)
    n = 40 if th == 1 else 300
    t0 = time.perf_counter()
    for t in range(n): ov.step_only(acts[t % 32], threads=th)
    print(os.environ.get("OMP_PLACES"), os.environ.get("OMP_PROC_BIND"), th, "threads:", f"{B * n / (time.perf_counter() - t0):.3e}", "env-steps/s")
