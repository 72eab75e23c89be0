"""bench_common.py -- what bench.py and bench_legs.py share: the workload's constants, the CPU-baseline leg, the oracle replay of a timed run and the
readers of the committed profiles.  (The oracle is imported here only inside cpu_baseline / oracle_replay: the checker and the reported CPU baseline.)"""
from __future__ import annotations

import os

# the CPU-baseline leg pins its OpenMP threads; the OpenMP runtime reads these when it is first loaded (and then binds the
# main thread to its first place, so the CPUs this process may use are counted before that)
NPROC = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def _cpu_quota():
    """CPUs this container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


CPU_QUOTA = _cpu_quota()
ORACLE_THREADS = NPROC if CPU_QUOTA is None else max(1, min(NPROC, int(CPU_QUOTA + 0.5)))  # never more threads than CPUs the container may run
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")  # one thread per physical core: with "threads" 16 threads share 8 cores' SMT siblings (3.2e7 against 5.3e7 env-steps/s)
# multi-process GPU work on this pool needs dmabuf IPC (hipIpcGetMemHandle fails otherwise): set before anything can initialise HIP -- the
# ranks may come from the driver's launcher, not from launch_ranks() below
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import csv  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

NUM_QUBITS = 16
ENVS_PER_GPU = 65536
SCRAMBLE = 256
CHUNK = 256  # most steps one hipGraph replay holds
RING = 16    # pre-sampled action buffers the steps cycle through (a policy rewrites ONE buffer per step)
OBS_WORDS = 2 * NUM_QUBITS  # packed observation: one 32-bit word per tableau row
ALGO_BYTES_PER_STEP = 160  # 128 B state read + 16 B touched rows + 16 B scalars
# what the one-step kernel has to move per env: two 16-byte row groups read and (at most) written back,
# action 4 R, depth 4 R + 4 W, bad mask 4 R + 4 W (written when it changes), reward 4 W, done / success 1 W each; the 8-byte gate
# entry comes from a 1.4 KB table that stays cache resident
NEEDED_BYTES_PER_STEP = 2 * 16 + 2 * 16 + 4 + 8 + 8 + 4 + 2
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
CONTROL_TIMEOUT_S = 90   # gloo control plane: a rank that never reaches a barrier costs its peers this long, not the driver's whole budget
CADENCE_BUDGET_S = 240   # N > 1: wall-clock budget of the optional collective-cadence legs; past it the line is printed without them
KERNEL = "qg::qm_step1_kernel<16, true, false"  # prefix: the trailing template arguments (feature flags, done list) vary by call site
PROFILE_DIR = os.path.join(ROOT, "profiles", "r05")


def build_gateset():
    from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map, line_edges

    kinds = ["H", "S", "Sdg", "SX", "SXdg", "CX", "CZ", "SWAP"]
    return gateset_from_coupling_map(line_edges(NUM_QUBITS, True), None, kinds)


def global_actions(seed: int, total_envs: int, num_actions: int) -> torch.Tensor:
    """The ring of pre-sampled action buffers of the WHOLE batch, [RING, total_envs] int32 on the host: a function of
    (seed, global env id) only, so every rank takes its slice of the same tensor."""
    gen = torch.Generator()
    gen.manual_seed(seed)
    return torch.randint(0, num_actions, (RING, total_envs), dtype=torch.int32, generator=gen)


def cpu_baseline(gateset, seed: int, budget_s: float = 2.0, repeats: int = 5):
    """Time the CPU oracle (a C port of the reference's scalar Rust path, one env object per env, OpenMP over envs like
    twisterl's rayon-over-clones) on this box's host cores: the configuration's own 65 536 envs, on ONE core and on ALL
    cores the process may run on, one pinned thread per physical core (OMP_PLACES=cores, OMP_PROC_BIND=close), median of `repeats` timed repeats each."""
    from oracle import OracleEnv, OracleVec

    nproc = NPROC
    B = ENVS_PER_GPU
    A = len(gateset)
    proto = OracleEnv("clifford", NUM_QUBITS, gateset, add_inverts=0, add_perms=0, track_solution=0, difficulty=SCRAMBLE)
    ov = OracleVec(proto, B)
    rng = np.random.default_rng(seed)
    ov.reset_with(rng.integers(0, A, size=(SCRAMBLE, B)))
    acts = rng.integers(0, A, size=(32, B)).astype(np.int32)

    def measure(threads: int):
        for t in range(4):  # warm-up: thread pool, first touch
            ov.step_only(acts[t], threads=threads)
        t0 = time.perf_counter()
        ov.step_only(acts[4], threads=threads)
        one = max(time.perf_counter() - t0, 1e-6)
        n_steps = int(max(4, min(100_000, budget_s / one)))
        for t in range(max(4, n_steps // 4)):  # untimed: the first passes after a thread-count change run slow
            ov.step_only(acts[t % 32], threads=threads)
        rates = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            for t in range(n_steps):
                ov.step_only(acts[t % 32], threads=threads)
            rates.append(B * n_steps / (time.perf_counter() - t0))
        return float(np.median(rates)), [float(r) for r in rates], n_steps

    quota, threads = CPU_QUOTA, ORACLE_THREADS
    one_core, one_core_runs, n1 = measure(1)
    all_core, all_core_runs, nall = measure(threads)
    return {
        "value": all_core,
        "unit": "env-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"CliffordGym 16q, {B} envs x {nall} steps per repeat, median of {repeats} repeats; C port of the reference scalar "
                  f"path (byte-per-entry state, per-env objects, gcc -O3), OpenMP static over envs, one thread per physical core (OMP_PLACES=cores, OMP_PROC_BIND=close)",
        "repeats": all_core_runs,
        "one_core": {"value": one_core, "cores": 1, "repeats": one_core_runs, "steps_per_repeat": n1},
        "nproc": nproc,
        "cpu_quota": quota,
    }


def oracle_replay(gateset, seed, global_ids, ring_actions, trace):
    """The run the GPU did -- same seed, same scramble draws (functions of the GLOBAL env id), same action buffers in the same
    order -- on the CPU oracle for the sampled envs.  Returns the oracle batch and the outputs of its last step."""
    from oracle import OracleEnv, OracleVec

    proto = OracleEnv("clifford", NUM_QUBITS, gateset, add_inverts=0, add_perms=0, track_solution=0, difficulty=SCRAMBLE)
    ov = OracleVec(proto, len(global_ids))
    ov.reset_seeded(seed, env_ids=global_ids, threads=ORACLE_THREADS)  # Env::reset with the draws of qg_vec_reset(seed) (counter RNG, global env id)
    last = (None, None, None, None)
    for ring_idx in trace:
        last = ov.step(ring_actions[ring_idx], threads=ORACLE_THREADS)
    return ov, last


def pack_rows_u32(dense: np.ndarray) -> np.ndarray:
    """dense [n, 32, 32] of {0,1} -> packed [n, 32] uint32, bit c of word r = entry (r, c) (QG_FMT_PACKED)."""
    w = (dense.astype(np.uint64) << np.arange(dense.shape[2], dtype=np.uint64)).sum(axis=2)
    return w.astype(np.uint32)


def parity_replay(gateset, seed, global_ids, ring_actions, trace, snap):
    """Compare everything env.step() produces (reward bits, success, is_final, depth, dense observation) after the last timed step."""
    import hashlib

    from util import f32_bits

    ov, (r, s, f, d) = oracle_replay(gateset, seed, global_ids, ring_actions, trace)
    n = len(global_ids)
    ok = {
        "reward_bits": bool(np.array_equal(f32_bits(snap["reward"]), f32_bits(r))),
        "success": bool(np.array_equal(snap["success"], s)),
        "is_final": bool(np.array_equal(snap["done"], f)),
        "depth": bool(np.array_equal(snap["depth"], d)),
        "observation": bool(np.array_equal(snap["obs"].reshape(n, -1), ov.observe_dense(threads=ORACLE_THREADS))),
    }

    def digest(obs, reward, success, depth):  # SHA-256 over the final (state, reward bits, success, depth) streams
        h = hashlib.sha256()
        for arr in (np.asarray(obs, dtype=np.uint8), f32_bits(reward).astype(np.uint32), np.asarray(success, dtype=np.uint8), np.asarray(depth, dtype=np.int32)):
            h.update(np.ascontiguousarray(arr).tobytes())
        return h.hexdigest()

    sha_gpu = digest(snap["obs"].reshape(n, -1), snap["reward"], snap["success"], snap["depth"])
    sha_cpu = digest(ov.observe_dense(threads=ORACLE_THREADS), r, s, d)
    ok["sha256"] = sha_gpu == sha_cpu
    return {"envs": int(n), "steps_replayed": len(trace), "checked": sorted(ok), "bit_exact": all(ok.values()),
            "mismatch": [k for k, v in ok.items() if not v], "sha256_hip": sha_gpu, "sha256_oracle": sha_cpu}


def gathered_parity(gateset, seed, global_ids, ring_actions, trace, shard):
    """The learner-side view: the sampled envs' slice of the all-gathered shard (packed observation words, reward, is_final, success)
    against the oracle replayed up to the step the snapshot was taken at."""
    from util import f32_bits

    ov, (r, s, f, _) = oracle_replay(gateset, seed, global_ids, ring_actions, trace)
    n = len(global_ids)
    want_obs = pack_rows_u32(ov.observe_dense(threads=ORACLE_THREADS).reshape(n, 2 * NUM_QUBITS, 2 * NUM_QUBITS))
    ok = {
        "packed_observation": bool(np.array_equal(shard["obs"].view(np.uint32), want_obs)),
        "reward_bits": bool(np.array_equal(f32_bits(shard["reward"]), f32_bits(r))),
        "is_final": bool(np.array_equal(shard["done"], f)),
        "success": bool(np.array_equal(shard["success"], s)),
    }
    return {"envs": int(n), "steps_replayed": len(trace), "checked": sorted(ok), "bit_exact": all(ok.values()),
            "mismatch": [k for k, v in ok.items() if not v]}


def rocprof_kernel_avg_us(envs: int, required: bool = False):
    """Average duration of the step kernel in the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/r05/, tools/profile_bench.sh).  The statistics hold one batch size (profiling runs pass --no-large-batch).  Every
    instantiation of the kernel whose name starts with KERNEL counts (calls-weighted).  `required`: a missing file or kernel is an
    error -- the line's roofline.frac is this figure -- unless the run IS the profiling run (--profiling-run)."""
    path = os.path.join(PROFILE_DIR, "bench_kernel_stats.csv" if envs == ENVS_PER_GPU else f"bench_{envs}_kernel_stats.csv")
    want = KERNEL.split("::", 1)[1]
    calls, total_ns, mins = 0, 0.0, []
    try:
        with open(path) as f:
            for row in csv.DictReader(f):
                if want in row["Name"]:
                    calls += int(row["Calls"])
                    total_ns += float(row["AverageNs"]) * int(row["Calls"])
                    mins.append(float(row["MinNs"]))
    except OSError:
        calls = 0
    if calls:
        return {"avg_us": total_ns / calls / 1e3, "min_us": min(mins) / 1e3, "calls": calls, "source": os.path.relpath(path, ROOT)}
    if required:
        raise SystemExit(f"bench.py: {os.path.relpath(path, ROOT)} does not hold a kernel named {want}*: re-run tools/profile_bench.sh on the "
                         "current build and commit profiles/r05 (or pass --profiling-run)")
    return None


def profiled_configs():
    """profiles/r05/traffic.json: per configuration the step kernel's rocprofv3 average, the PMC bytes per env (separate FETCH_SIZE /
    WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), the bytes it needs and both fractions of 8 TB/s."""
    try:
        return json.load(open(os.path.join(PROFILE_DIR, "traffic.json")))["configs"]
    except Exception:
        return {}


def pmc_traffic(envs: int):
    """HBM bytes per launch of the step kernel from the committed PMC passes (tools/profile_bench.sh + tools/pmc_traffic.py:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), or None."""
    path = os.path.join(PROFILE_DIR, "traffic.json")
    try:
        t = json.load(open(path))
        e = t["by_envs"][str(envs)]
        return {"bytes_per_launch": e["bytes_per_launch"], "fetch_bytes": e["fetch_bytes"], "write_bytes": e["write_bytes"],
                "source": os.path.relpath(path, ROOT)}
    except Exception:
        return None
