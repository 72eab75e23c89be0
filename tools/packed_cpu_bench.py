""""also time the packed CPU backend for context".  CliffordGym 16q x 65 536 envs (the headline workload) stepped on the host on
bit-packed rows (tools/packed_cpu_bench.c, OpenMP over envs), checked against the oracle first (same states, reward bits, flags after 32
steps), then timed on one core and on all cores next to the oracle -- the byte-per-entry port of the reference that bench.py reports as
`cpu_baseline`.  Development tool: nothing in qiskit_gym_amd/ uses it."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import OracleEnv, OracleVec  # noqa: E402
from util import f32_bits, line_gateset  # noqa: E402

N, B, SCRAMBLE = 16, 65536, 256
so = os.path.join(ROOT, "tools", "bin", "libpacked_cpu.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", os.path.join(ROOT, "tools", "packed_cpu_bench.c"), "-o", so], check=True)
L = C.CDLL(so)


class Gate(C.Structure):
    _fields_ = [("type", C.c_uint8 * 2), ("dst", C.c_uint8 * 2), ("src", C.c_uint8 * 2), ("pad", C.c_uint8 * 2), ("penalty", C.c_float)]


def gate_table(gateset):
    """clifford.rs:89-133 as row operations; penalties of the default weights (metrics.rs:157-166) in the reference's f32 order."""
    w_c, w_g = np.float32(0.01), np.float32(0.0001)
    tab = (Gate * len(gateset))()
    for i, (name, qs) in enumerate(gateset):
        k = name.lower()
        a = qs[0]
        b = qs[1] if len(qs) > 1 else 0
        ops, dc, dg = [], 0, 1
        if k == "h": ops = [(2, a, N + a)]
        elif k in ("s", "sdg"): ops = [(1, N + a, a)]
        elif k in ("sx", "sxdg"): ops = [(1, a, N + a)]
        elif k == "cx": ops, dc, dg = [(1, b, a), (1, N + a, N + b)], 1, 1
        elif k == "cz": ops, dc, dg = [(1, N + a, b), (1, N + b, a)], 1, 3
        elif k == "swap": ops, dc, dg = [(2, a, b), (2, N + a, N + b)], 3, 3
        for j, (t, d, s) in enumerate(ops):
            tab[i].type[j], tab[i].dst[j], tab[i].src[j] = t, d, s
        tab[i].penalty = float(np.float32(np.float32(np.float32(w_c * np.float32(dc)) + np.float32(0.0)) + np.float32(0.0)) + np.float32(w_g * np.float32(dg)))
    return tab


gs = line_gateset("clifford", N)
A = len(gs)
tab = gate_table(gs)
proto = OracleEnv("clifford", N, gs, add_inverts=0, add_perms=0, track_solution=0, difficulty=SCRAMBLE)
ov = OracleVec(proto, B)
rng = np.random.default_rng(0x5EED0003)
ov.reset_with(rng.integers(0, A, size=(SCRAMBLE, B)))
dense = ov.observe_dense().reshape(B, 2 * N, 2 * N).astype(np.uint64)
rows = np.ascontiguousarray((dense << np.arange(2 * N, dtype=np.uint64)).sum(axis=2).astype(np.uint32))
depth = np.full(B, 128, dtype=np.int32)  # min(depth_slope * difficulty, max_depth)
reward = np.zeros(B, dtype=np.float32)
done = np.zeros(B, dtype=np.uint8)
success = np.zeros(B, dtype=np.uint8)
acts = rng.integers(0, A, size=(32, B)).astype(np.int32)


def pk_step(a, threads):
    L.pk_step(rows.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p), reward.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p),
              success.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), tab, A, C.c_int64(B), 2 * N, threads)


for t in range(32):  # the check: everything env.step() produces, after every step
    r, s, f, d = ov.step(acts[t])
    pk_step(acts[t], 0)
    assert np.array_equal(f32_bits(reward), f32_bits(r)) and np.array_equal(success, s) and np.array_equal(done, f) and np.array_equal(depth, d), t
dense = ov.observe_dense().reshape(B, 2 * N, 2 * N).astype(np.uint64)
assert np.array_equal(rows, (dense << np.arange(2 * N, dtype=np.uint64)).sum(axis=2).astype(np.uint32))
print("packed CPU model == oracle after 32 steps (states, reward bits, success, is_final, depth)")

nproc = len(os.sched_getaffinity(0))
try:  # never more threads than the container may run at once (cgroup v2 cpu.max), as bench.py's cpu_baseline
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
        nproc = max(1, min(nproc, int(float(q) / float(per) + 0.5)))
except Exception:
    pass


def rate(fn, threads, budget=2.0):
    fn(acts[0], threads)
    t0 = time.perf_counter()
    fn(acts[1], threads)
    one = max(time.perf_counter() - t0, 1e-6)
    n = int(max(4, min(20000, budget / one)))
    t0 = time.perf_counter()
    for i in range(n):
        fn(acts[i % 32], threads)
    return B * n / (time.perf_counter() - t0)


for threads in (1, nproc):
    pk = rate(pk_step, threads)
    orc = rate(lambda a, th: ov.step_only(a, threads=th), threads)
    print(f"{threads:3d} thread(s): packed rows {pk:.3e} env-steps/s   byte-per-entry port (oracle) {orc:.3e}   ratio {pk / orc:.1f}")
