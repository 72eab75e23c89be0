#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched env.step() hot path on MI355X.

Workload (BASELINE.json metric): CliffordGym, 16 qubits, line-16
bidirectional coupling map, all 8 gate kinds (170 actions), 65 536 envs per GPU, start = identity
scrambled by 256 uniform random actions, then uniform random actions, add_inverts=False,
add_perms=False, track_solution=False, default metric weights, free-running (no reset inside the
timed loop).  A "step" is ONE env.step() of every env = one step-kernel launch that gathers the rows
its env's action touches from the packed tableau in device memory, applies the gate, updates the
solved test, writes reward / done / success / depth and the touched rows back.  Inputs (actions) are
resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU; the data path is libqgym's own communicator (qg_comm_* of include/qgym.h: ncclAllGather of librccl.so.1
called from C++, and the direct write into the peers' windows over xGMI) -- torch.distributed (gloo, CPU) only carries the control
plane: the 128-byte communicator id, the barriers and the max-over-ranks of the elapsed time.  Started by the driver through torch.distributed.run (WORLD_SIZE set), or, when
it is not, bench.py itself starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
process before this process has touched a GPU, and relays rank 0's JSON line.  Fewer visible GPUs than N is an
error.  Rank r owns envs [r * 65 536, (r + 1) * 65 536) of ONE batch of N * 65 536 envs (weak scaling): every
counter-RNG draw and every action is a function of the global env id (qg_vec_set_env_base), so the sharded run
is bit-identical to the unsharded one.  env.step needs no collective.  The exchange BASELINE.json's north_star
names -- the observation handed back to the learner -- is an all-gather of one flat shard per rank (bit-packed
observation 8 MiB + rewards + is_final / success flags) on a side stream, double buffered and overlapped with
the following steps, once per rollout segment of min(--gather-every, K) steps, so the timed region always
contains at least one collective; the per-step cadence (SURVEY.md 8e's literal "one all-gather per step",
link-bound at >= 55 us against a 3 us step) is measured beside it and printed too.

Prints one JSON line on rank 0.
"""
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

import argparse  # noqa: E402
import csv  # noqa: E402
import json  # noqa: E402
import socket  # noqa: E402
import subprocess  # noqa: E402
import threading  # noqa: E402
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


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    """Start n ranks of this script under torch.distributed.run as a CHILD process (subprocess.run) and relay rank 0's JSON line.  This
    parent may already have touched the HIP runtime -- counting devices can initialise it -- which is exactly why the ranks are a fresh child
    and never an exec of this process (a process that has touched the GPU must not be replaced by another program)."""
    visible = torch.cuda.device_count()
    if visible < n and "--ranks-share-gpu0" not in argv:
        print(f"bench.py: --gpus {n} requested but only {visible} GPU(s) are visible; refusing to run fewer ranks and report them as {n}",
              file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in res.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
    if res.returncode != 0 or line is None:
        sys.stderr.write(res.stdout[-4000:])
        print(f"bench.py: the {n}-rank child run failed (exit code {res.returncode})", file=sys.stderr)
        return res.returncode or 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the CPU-oracle replay of the timed run")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the 2^18 / 2^20 / 2^22-env legs (profiling runs: keeps the kernel statistics to one batch size)")
    ap.add_argument("--no-default-config", action="store_true", help="skip the reference-default (add_inverts=True, track_solution=True) leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 / C5 / C3d legs (SURVEY 8d's other configurations)")
    ap.add_argument("--profiling-run", action="store_true",
                    help="this run produces profiles/r05 (tools/profile_bench.sh): the committed rocprofv3 summary is not required and roofline.frac falls "
                         "back to the live clock")
    ap.add_argument("--no-dense-obs", action="store_true", help="skip the observation-mode legs (SURVEY 8d: packed and dense observation after every step)")
    ap.add_argument("--no-collector", action="store_true", help="skip the policy-in-the-loop leg (SURVEY 8f-3: collection with the reference's default policy shape)")
    ap.add_argument("--no-gather", action="store_true", help="N>1 diagnostics: step only, no all-gather")
    ap.add_argument("--no-p2p", action="store_true", help="N>1: skip the direct-write (hipIpc windows over xGMI) cadence leg")
    ap.add_argument("--handover", choices=["rccl", "direct"], default="rccl",
                    help="N>1: transport of the learner shard inside the timed region: ncclAllGather called from libqgym on a side stream (default), or the "
                         "direct write into every rank's hipIpc window (no RCCL at all: communicator handles travel over the control plane)")
    ap.add_argument("--ranks-share-gpu0", action="store_true",
                    help="diagnostics: every rank uses GPU 0 (a whole N-rank job -- control plane, sharding, hand-over, parity -- on a one-GPU box; needs "
                         "--handover direct, RCCL refuses two ranks on one GPU)")
    ap.add_argument("--gather-every", type=int, default=CHUNK,
                    help="N>1: all-gather the learner shard every min(this, --steps) steps (1 = after every step)")
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU,
                    help="diagnostics: envs per GPU (the metric is quoted at the default 65 536; profiles/ uses larger batches to show where the "
                         "launch boundary stops mattering)")
    ap.add_argument("--force-multi", action="store_true",
                    help="diagnostics: run the multi-GPU code path (RCCL init, side-stream all-gather) on ONE rank")
    ap.add_argument("--shard", type=str, default=None,
                    help="with --force-multi: R/W = be rank R of a W-rank job for every env id (env base R * envs, actions of that slice)")
    ap.add_argument("--inject-p2p-open-failure", type=int, default=-1, metavar="RANK",
                    help="tests only: this rank's hipIpcOpenMemHandle phase raises (tests/test_gpu_multi.py: every rank must stop at the same phase)")
    ap.add_argument("--diag-repeat", type=int, default=0, help="diagnostics: repeat the timed region this many times after the measurement and print each on stderr")
    ap.add_argument("--dump-gathered", type=str, default=None,
                    help="rank 0: write a strided sample of this rank's part of the last all-gathered shard, with what an oracle replay needs, to this .npz")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.gather_every < 1:
        ap.error("--gpus, --steps, --gather-every must be positive and --warmup non-negative")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1 and not args.force_multi:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(world_env) if world_env is not None else 1
    if world != args.gpus and not args.force_multi:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)

    # stdout carries exactly one JSON line: libraries that print to the C-level stdout (RCCL's version
    # banner) are sent to stderr, our line goes to a private duplicate of the original descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.ranks_share_gpu0 and args.handover != "direct":
        ap.error("--ranks-share-gpu0 needs --handover direct")
    from qiskit_gym_amd.distributed import device_for_rank

    try:  # GPU LOCAL_RANK (or GPU 0 for every rank of a rehearsal on a one-GPU box): tests/test_distributed_cpu.py
        local_rank = device_for_rank(local_rank, int(world_env or 1), torch.cuda.device_count(), args.ranks_share_gpu0)
    except RuntimeError as exc:
        print(f"bench.py: rank {rank}: {exc}", file=sys.stderr)
        sys.exit(2)
    dist = None
    comm = None
    multi = world > 1 or args.force_multi
    if multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the control plane stays on loopback (the hostname may not resolve)
        torch.cuda.set_device(local_rank)
        # control plane only (CPU tensors): id broadcast, barriers, max-over-ranks of the elapsed time
        import datetime

        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=CONTROL_TIMEOUT_S))
        if dist.get_world_size() != world:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, expected {world}")
        from qiskit_gym_amd.distributed import Communicator

        if args.handover == "rccl":
            uid = [Communicator.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = Communicator(rank, world, device=local_rank, unique_id=uid[0])  # ncclCommInitRank inside libqgym
        else:
            comm = Communicator(rank, world, device=local_rank, local=True)  # windows are connected below, once the shard size is known
    else:
        torch.cuda.set_device(0)
    n_gpus = world
    dev = torch.device("cuda", torch.cuda.current_device())

    from qiskit_gym_amd.distributed import shard_range
    from qiskit_gym_amd.vec import VecEnv

    n, gateset = build_gateset()
    A = len(gateset)
    B = args.envs
    # which slice of which batch this rank steps: rank r of W owns envs [r * B, (r + 1) * B) of a batch of W * B
    shard_rank, shard_world = rank, world
    if args.shard:
        if not args.force_multi or world != 1:
            raise SystemExit("bench.py: --shard needs --force-multi on one rank")
        shard_rank, shard_world = (int(x) for x in args.shard.split("/"))
    total_envs = B * shard_world
    env_base, count = shard_range(total_envs, shard_rank, shard_world)
    assert count == B
    seed = 0x5EED0003 if shard_world == 1 else 0x5EED0004  # seed = 0x5EED0000 + config (3: one GPU, 4: the sharded batch)
    env = VecEnv("clifford", n, gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE, env_base=env_base)
    stream = torch.cuda.Stream(device=dev)
    K, W = args.steps, args.warmup

    host_actions = global_actions(seed, total_envs, A)[:, env_base:env_base + B].contiguous()
    actions = host_actions.to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 17)

    ring_trace = []  # which action buffer every step since the last reset used (for the oracle replay)
    launches = []    # how the steps since the last reset were issued (config.launch)

    def issue(nsteps: int, eager: bool = False):
        """nsteps env.step() launches of the resident batch: chunks of <= CHUNK single-step launches, each chunk one replay of a cached
        hipGraph (the same launches, results and memory traffic as that many qg_vec_step calls, without their host cost).  `eager`: plain
        qg_vec_step launches -- the untimed warm-up steps use it so that the timed region's graph is the graph launched last (launching
        another graph executable in between costs the next launch ~13 us, tools/sync_probe.py)."""
        done = 0
        while done < nsteps:
            c = 1 if eager else min(CHUNK, nsteps - done)
            if c >= 2:
                env.rollout_ring(actions, c)
                launches.append(("graph", c))
            else:
                env.step(actions[done % RING] if eager else actions[0])
                launches.append(("eager", 1))
            ring_trace.extend(((done + i) % RING) if eager else (i % RING) for i in range(c))
            done += c

    # ---- multi-GPU: step + all-gather of the learner shard, double buffered -------------------
    gather_every = min(args.gather_every, K)
    gather_log = {"submitted": 0}
    if multi:
        from qiskit_gym_amd.distributed import run_guarded_phases, split_gathered

        # double-buffered, host-mediated hand-over to the communicator's side stream, inside libqgym (qg_comm_gather_submit: a
        # stream-to-stream event wait would slow every later graph replay on the step stream by ~40 %).  One flat shard per rank
        # carries everything SURVEY 8e lists for the learner: the packed observation [B, 32], the f32 rewards [B] and is_final /
        # success [B] bytes each -- one ncclAllGather instead of four
        layout = env.shard_layout()

        direct = args.handover == "direct"
        p2p_state = {"connected": False}

        def connect_windows():
            """hipIpc handles over the control plane (gloo), then every rank maps every other rank's window.  Each phase is local to a rank
            and followed by a vote of all ranks: a rank whose hipIpcOpenMemHandle fails stops everybody at the same point (None = connected)."""
            box = {}

            def export():
                box["mine"] = comm.p2p_export(int(layout.bytes))

            def exchange():
                box["handles"] = [None] * world
                dist.all_gather_object(box["handles"], box["mine"])

            def open_():
                if args.inject_p2p_open_failure == rank:  # tests: this rank cannot map its peers
                    raise RuntimeError("injected: hipIpcOpenMemHandle failed")
                comm.p2p_open(box["handles"])

            err = run_guarded_phases([("p2p_export", export), ("handle_exchange", exchange), ("p2p_open", open_)])
            p2p_state["connected"] = err is None
            return err

        if direct:
            err = connect_windows()
            if err is not None:  # the transport the caller asked for does not exist: no headline number in this mode
                print(f"bench.py: rank {rank}: --handover direct cannot be set up: {err}", file=sys.stderr)
                sys.exit(3)
        direct_latest = [None]

        def snapshot_and_gather():
            if direct:  # pack + one copy kernel into all windows + flags, on the step stream; the view is copied out before the window is released
                comm.push(env)
                direct_latest[0] = comm.wait().clone()
                comm.release()
            else:
                comm.submit(env)
            gather_log["submitted"] += 1
            gather_log["trace_len"] = len(ring_trace)

        class _Flush:
            flush = staticmethod(lambda: None if direct else comm.flush())
            latest = staticmethod(lambda: direct_latest[0] if direct else comm.latest())

        gatherer = _Flush

        def run_steps(nsteps: int, eager: bool = False):
            """Each rank steps its own shard (no collective inside step).  Every `gather_every` steps the learner shard is snapshotted
            and all-gathered on the side stream, double buffered, overlapping the steps that follow."""
            done = 0
            while done < nsteps:
                c = min(gather_every, nsteps - done)
                issue(c, eager)
                done += c
                if not args.no_gather and c == gather_every:
                    snapshot_and_gather()
            if not args.no_gather:
                gatherer.flush()
    else:
        run_steps = issue

    def timed_region():
        """EXACTLY K steps between two device-wide synchronisations (the caller has synchronised before): wall clock + HIP events on the launch stream."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_start = time.perf_counter()
        with torch.cuda.stream(stream):
            e0.record(stream)
            run_steps(K)
            e1.record(stream)
        torch.cuda.synchronize()
        return time.perf_counter() - t_start, e0, e1

    with torch.cuda.stream(stream):
        env.reset(seed)
        run_steps(K)  # builds and caches every graph the timed region replays (setup, not a step)
    torch.cuda.synchronize()
    timed_region()  # dress rehearsal (setup): the first pass through this host code and the graph's second launch pay one-time costs
    with torch.cuda.stream(stream):
        if W:
            run_steps(W)
        if multi and not args.no_gather:  # communicator set-up and first-use kernel loads of RCCL (setup, not a step)
            snapshot_and_gather()
            snapshot_and_gather()
            gatherer.flush()
        env.reset(seed)
        ring_trace.clear()
        launches.clear()
        gather_log["submitted"] = 0
        if W:
            run_steps(W, True)  # untimed warmup steps, as plain launches (see issue())
        launches.clear()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gathers_before = gather_log["submitted"]
    elapsed, ev0, ev1 = timed_region()
    if args.diag_repeat:  # diagnostics only: the same timed region again, to tell one-shot effects from steady state
        for _ in range(args.diag_repeat):
            td = time.perf_counter()
            with torch.cuda.stream(stream):
                run_steps(K)
            torch.cuda.synchronize()
            print(f"diag repeat: {(time.perf_counter() - td) * 1e6:.1f} us for {K} steps", file=sys.stderr)
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    env.sync()  # raises if any env faulted
    stream_ms = ev0.elapsed_time(ev1)
    timed_launches = list(launches)
    gathers_timed = gather_log["submitted"] - gathers_before

    # ---- snapshot of EVERY env right after the timed steps, for the oracle replay (every lane of every wave: the kernels are
    # lane-position sensitive, and the oracle steps 65 536 envs x a few thousand steps in seconds) -------
    snap = gshard = None
    ids = np.arange(B)
    idx = torch.as_tensor(ids, device=dev)
    if rank == 0 and not args.no_parity:
        with torch.cuda.stream(stream):
            snap = {
                "obs": env.observe()[idx].cpu().numpy(),
                "reward": env.reward[idx].cpu().numpy(),
                "success": env.success[idx].cpu().numpy(),
                "done": env.done[idx].cpu().numpy(),
                "depth": env.depth[idx].cpu().numpy(),
            }
        snap_trace = list(ring_trace)
    snap_actions = host_actions[:, ids].numpy()
    if multi and rank == 0 and not args.no_gather and gatherer.latest() is not None:
        # this rank's own part of the last all-gathered buffer, as the learner would read it
        g_obs, g_rew, g_done, g_succ = split_gathered(gatherer.latest(), layout, world, 4)
        lo = rank * B
        gshard = {"obs": g_obs[lo:lo + B][idx].cpu().numpy(), "reward": g_rew[lo:lo + B][idx].cpu().numpy(),
                  "done": g_done[lo:lo + B][idx].cpu().numpy(), "success": g_succ[lo:lo + B][idx].cpu().numpy(),
                  "trace": ring_trace[:gather_log["trace_len"]]}
        if world > 1:  # ... and the LAST rank's part, which only the hand-over can have brought here (its envs, its slice of the actions)
            plo = (world - 1) * B
            gshard["peer"] = {"obs": g_obs[plo:plo + B][idx].cpu().numpy(), "reward": g_rew[plo:plo + B][idx].cpu().numpy(),
                              "done": g_done[plo:plo + B][idx].cpu().numpy(), "success": g_succ[plo:plo + B][idx].cpu().numpy(),
                              "trace": gshard["trace"], "global_ids": plo + ids,
                              "actions": global_actions(seed, total_envs, A)[:, plo:plo + B][:, ids].numpy()}
        if args.dump_gathered:
            np.savez(args.dump_gathered, global_ids=env_base + ids, actions=snap_actions, seed=np.uint64(seed), scramble=SCRAMBLE,
                     trace=np.asarray(gshard["trace"], dtype=np.int64), obs=gshard["obs"], reward=gshard["reward"], done=gshard["done"],
                     success=gshard["success"], total_envs=total_envs, env_base=env_base)

    # ---- collective cadences, beside the timed region (N > 1 or --force-multi): run LAST, after everything the line needs has been
    # measured, every phase followed by a vote of all ranks (run_guarded_phases) and the whole block under a wall-clock budget -- a rank that
    # fails (or a peer that never shows up) costs these optional figures, never the headline line -------------------
    if multi and not args.no_gather and args.handover == "direct":
        with torch.cuda.stream(stream):
            comm.check()  # a peer that missed a deadline during the run is an error, not a slow number

    def collective_cadences():
        if not (multi and not args.no_gather and args.handover == "rccl"):
            return None

        def timed(fn, reps):
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t = time.perf_counter()
            with torch.cuda.stream(stream):
                for _ in range(reps):
                    fn()
                gatherer.flush()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()) / reps

        def step_and_gather():
            env.step(actions[0])
            snapshot_and_gather()

        cadence = {"segment_steps": gather_every, "shard_bytes_per_rank": int(layout.bytes)}
        inline_out = torch.empty(int(layout.bytes) * world, dtype=torch.uint8, device=dev)

        def step_and_gather_in_stream():  # qg_vec_gather_learner_shard: pack + ncclAllGather on the step stream itself
            env.step(actions[0])
            comm.gather(env, out=inline_out)

        def rccl_legs():  # medians of three batches each
            per_step = float(np.median([timed(step_and_gather, 32) for _ in range(3)]))       # SURVEY 8e's literal cadence: one all-gather per env.step()
            gather_only = float(np.median([timed(snapshot_and_gather, 16) for _ in range(3)]))  # snapshot + collective alone, nothing to overlap with
            in_stream = float(np.median([timed(step_and_gather_in_stream, 32) for _ in range(3)]))
            cadence.update({"per_step_gather_us": per_step * 1e6, "per_step_gather_value": B * n_gpus / per_step,
                            "in_stream_gather_us": in_stream * 1e6, "segment_gather_us": gather_only * 1e6})

        err = run_guarded_phases([("rccl_cadences", rccl_legs)])
        if err is not None:
            cadence["error"] = err
            return cadence
        # the direct write (SURVEY 5's follow-up): every rank copies its shard into a window in each peer's HBM over xGMI and raises
        # a flag there; no collective library on the data path.  Measured beside the RCCL cadences, and checked against them.
        if not args.no_p2p:
            err = connect_windows()
            if err is None:
                p2p_view = [None]
                res = {}

                def step_push_wait():
                    env.step(actions[0])
                    comm.push(env)
                    p2p_view[0] = comm.wait()
                    comm.release()

                def push_wait():
                    comm.push(env)
                    p2p_view[0] = comm.wait()
                    comm.release()

                def first_epoch():  # first use, checked at once: a peer that never arrives costs one deadline, not one per repetition
                    with torch.cuda.stream(stream):
                        step_push_wait()
                        comm.check()

                def timed_legs():  # median of three batches: now and then a launch stalls for tens of ms (as in the collector leg)
                    timed(step_push_wait, 4)
                    res["p2p_step"] = float(np.median([timed(step_push_wait, 32) for _ in range(3)]))
                    res["p2p_only"] = float(np.median([timed(push_wait, 16) for _ in range(3)]))

                def compare():
                    with torch.cuda.stream(stream):
                        comm.push(env)
                        view = comm.wait()
                        direct_copy = view.clone()
                        comm.release()
                        comm.gather(env, out=inline_out)
                        comm.check()
                    res["same"] = bool(torch.equal(direct_copy, inline_out))
                    if not res["same"]:
                        raise RuntimeError("the direct-write hand-over and the RCCL all-gather disagree")

                err = run_guarded_phases([("first_epoch", first_epoch), ("timed", timed_legs), ("equals_rccl_gather", compare)])
                if err is None:
                    cadence["direct_write"] = {"per_step_us": res["p2p_step"] * 1e6, "per_step_value": B * n_gpus / res["p2p_step"],
                                               "push_wait_release_us": res["p2p_only"] * 1e6, "equals_rccl_gather": res["same"],
                                               "what": "qg_vec_push_learner_shard + qg_comm_p2p_wait + qg_comm_p2p_release after every env.step(): the pack, "
                                                       "one copy kernel writing into all ranks' hipIpc-mapped windows, per-source arrival flags"}
            if err is not None:  # reported, not fatal: the RCCL path above is the measured default
                cadence["direct_write"] = {"error": err}
        return cadence

    # ---- roofline leg: duration of the step kernel, HIP events on the stream the kernel is launched on --------
    # (1) the timed region itself: events around the K timed launches (what `achieved` uses)
    timed_region_us = stream_ms * 1e3 / K
    # (2) steady-state launch period inside one hipGraph of CHUNK launches (no host in the loop)
    with torch.cuda.stream(stream):
        env.rollout_ring(actions, CHUNK)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record(stream)
        for _ in range(4):
            env.rollout_ring(actions, CHUNK)
        g1.record(stream)
    torch.cuda.synchronize()
    graph_period_us = g0.elapsed_time(g1) * 1e3 / (4 * CHUNK)
    # (3) one eager launch between two events
    reps = 200
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    with torch.cuda.stream(stream):
        for i in range(reps):
            starts[i].record(stream)
            env.step(actions[i % RING])
            stops[i].record(stream)
    torch.cuda.synchronize()
    eager_event_us = float(np.median(sorted(s.elapsed_time(e) * 1e3 for s, e in zip(starts, stops))))
    # (4) the kernel's own clock (qg_vec_set_kernel_clock): every wave stamps its entry and -- after waiting for its loads and stores -- its exit on
    # the device's constant-rate counter; a launch's duration = last exit - first entry.  No host, no profiler, no launch boundary: this is the
    # duration roofline.achieved divides by.  One slot per launch of a CHUNK-launch graph, four replays.
    kslots = env.kernel_clock(CHUNK)
    kdurs = []
    with torch.cuda.stream(stream):
        env.rollout_ring(actions, CHUNK)  # a fresh graph: its launches carry their slots
        for _ in range(4):
            torch.cuda.synchronize()
            kslots.zero_()
            torch.cuda.synchronize()
            env.rollout_ring(actions, CHUNK)
            torch.cuda.synchronize()
            kdurs.append(env.kernel_durations_us(kslots))
    env.kernel_clock(0)
    kdurs = np.concatenate(kdurs)
    if kdurs.size != 4 * CHUNK:
        raise SystemExit(f"bench.py: {kdurs.size} of {4 * CHUNK} step launches stamped the kernel clock")
    device_clock_us = float(kdurs.mean())
    algo_bytes = ALGO_BYTES_PER_STEP * B
    needed_bytes = NEEDED_BYTES_PER_STEP * B
    achieved = algo_bytes / (device_clock_us * 1e-6) / 1e9
    rocprof = rocprof_kernel_avg_us(B)  # the committed rocprofv3 summary of this command: reported beside the live clock, never required
    traffic = pmc_traffic(B)

    # ---- fused rollout (state in LDS across steps), reported beside the headline --------
    fused = None
    if not multi:
        FT = 128
        facts = torch.randint(0, A, (FT, B), dtype=torch.int32, device=dev, generator=gen)
        with torch.cuda.stream(stream):
            env.rollout(facts, fused=True)
            torch.cuda.synchronize()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(stream)
            for _ in range(8):
                env.rollout(facts, fused=True)
            f1.record(stream)
        torch.cuda.synchronize()
        fms = f0.elapsed_time(f1)
        fused = {"value": B * FT * 8 / (fms * 1e-3), "unit": "env-steps/s", "steps_per_launch": FT,
                 "kernel": "qg::qm_fused_lds_kernel<16, true, false> (rows resident in LDS; actions known up front)"}

    def graph_period(venv, acts, replays=2):
        with torch.cuda.stream(stream):
            venv.rollout_ring(acts, CHUNK)
            torch.cuda.synchronize()
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record(stream)
            for _ in range(replays):
                venv.rollout_ring(acts, CHUNK)
            b1.record(stream)
        torch.cuda.synchronize()
        venv.sync()
        return b0.elapsed_time(b1) * 1e3 / (replays * CHUNK)

    # ---- the reference's DEFAULT configuration (envs/synthesis.py:182-204: add_inverts=True, track_solution=True) -------
    default_cfg = None
    if not multi and B == ENVS_PER_GPU and not args.no_default_config:
        DT = 128  # = max_depth (envs/synthesis.py:188): what one episode, and its solution log, can hold
        denv = VecEnv("clifford", n, gateset, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=SCRAMBLE)
        coins = torch.randint(0, 2, (DT, B), dtype=torch.uint8, device=dev, generator=gen)
        dacts = torch.randint(0, A, (DT, B), dtype=torch.int32, device=dev, generator=gen)
        dms = 0.0
        with torch.cuda.stream(stream):
            denv.reset(seed)
            denv.rollout(dacts, coins=coins)  # builds the graph
            for _ in range(4):
                denv.reset(seed)  # a new episode: the log is empty again (not timed)
                d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                d0.record(stream)
                denv.rollout(dacts, coins=coins)
                d1.record(stream)
                torch.cuda.synchronize()
                dms += d0.elapsed_time(d1)
        denv.sync()
        dus = dms * 1e3 / (4 * DT)
        default_cfg = {"us_per_step": dus, "value": B / (dus * 1e-6), "unit": "env-steps/s",
                       "config": "add_inverts=True (coin per env per step), track_solution=True, otherwise the headline workload; "
                                 f"episodes of {DT} steps, each one hipGraph of {DT} launches, coins given"}
        del denv, coins, dacts

    # ---- SURVEY 8(d)'s auto-reset variant of the headline workload: qg_vec_reset_done after every step (finished episodes start over on the
    # device: scramble from the identity), one captured graph of 128 x (step, reset_done).  Two episode schedules: "desynchronised" -- what a
    # collector sees: episode ends spread evenly over time, 1/128 of the batch finishes in every step (Env::reset called for class
    # env % 128 == k at warm-up step k) -- and "synchronised" (every env finishes in the same step, 127 of 128 reset_done calls find nothing) -----
    auto_reset = None
    if not multi and B == ENVS_PER_GPU and not args.no_default_config:
        AT = 128  # = min(depth_slope * difficulty, max_depth) for every configuration below: the episode length

        def auto_reset_leg(aenv, num_actions, spread=True):
            """One captured graph of AT x (step, reset_done), the pair issued as qg_vec_reset_done_step (ONE launch where the layout has it: TILE without
            add_inverts, LF8, PERM), replayed 4 times.  `spread`: Env::reset for class env % AT == k at warm-up step k, so 1 / AT of the batch finishes in
            every step of the graph (what a collector sees); else every env finishes in the same step."""
            nb = aenv.batch
            aacts = torch.randint(0, num_actions, (AT, nb), dtype=torch.int32, device=dev, generator=gen)
            afin = torch.empty((AT, nb), dtype=torch.uint8, device=dev)
            with torch.cuda.stream(stream):
                aenv.reset(seed)
                if spread:
                    cls = torch.arange(nb, device=dev) % AT
                    for k in range(AT):  # eager warm-up: spreads the episode ends (the done flags are caller-owned memory, qg_vec_bind_outputs)
                        aenv.set_counters(k, k)
                        aenv.step(aacts[k])
                        aenv.reset_done(seed + 0x51ED * (k + 1))
                        aenv.done[cls == k] = 1
                        aenv.reset_done(seed + 0xA5A5 * (k + 1))

                def episode():  # step, then AT - 1 x (reset_done, step) as qg_vec_reset_done_step, and the last reset_done
                    aenv.set_counters(0, 0)
                    aenv.rollout(aacts[0:1], dones_out=afin[0:1])
                    for t in range(1, AT):
                        aenv.set_counters(t, t)
                        aenv.reset_done_step(seed + 0x9E3779B9 * t, aacts[t], dones_out=afin[t])
                    aenv.reset_done(seed + 0x9E3779B9 * AT)

                episode()  # eager pass (allocations, kernel loads)
                torch.cuda.synchronize()
                ag = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ag, stream=stream):
                    episode()
                torch.cuda.synchronize()
                ag.replay()
                torch.cuda.synchronize()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(stream)
                for _ in range(4):
                    ag.replay()
                a1.record(stream)
            torch.cuda.synchronize()
            aenv.sync()
            aus = a0.elapsed_time(a1) * 1e3 / (4 * AT)
            per_step = afin.float().mean(dim=1)
            return {"us_per_step": aus, "value": nb / (aus * 1e-6), "unit": "env-steps/s", "envs": nb, "finished_per_step": float(per_step.mean()),
                    "finished_per_step_min_max": [float(per_step.min()), float(per_step.max())]}

        from util import line_gateset as _line_gateset  # (the gateset builder the configs leg below uses)

        legs = {}
        for schedule in ("desynchronised", "synchronised", "desynchronised_reference_defaults"):
            ref_defaults = schedule.endswith("reference_defaults")  # add_inverts + solution log: the pair is two launches behind the one call
            aenv = VecEnv("clifford", n, gateset, B, add_inverts=ref_defaults, add_perms=False, track_solution=ref_defaults, difficulty=SCRAMBLE)
            legs[schedule] = auto_reset_leg(aenv, A, spread=schedule != "synchronised")
            del aenv
        # "an auto-reset variant reported separately" for the other configurations too, same schedule (episode ends spread evenly over time)
        gs2a = _line_gateset("linear_function", 8)
        for name, nb, kw in (("C2", 8192, dict(add_inverts=False, track_solution=False)), ("C2_x65536", B, dict(add_inverts=False, track_solution=False)),
                             ("C2_reference_defaults", 8192, dict(add_inverts=True, track_solution=True))):
            aenv = VecEnv("linear_function", 8, gs2a, nb, add_perms=False, difficulty=64, **kw)
            legs[name] = dict(auto_reset_leg(aenv, len(gs2a)), config=f"LinearFunctionGym 8q x {nb}, difficulty 64, {kw}: word_reset_step_kernel (one launch per pair)")
            del aenv
        gs5a = _line_gateset("pauli", 20)
        aenv = VecEnv("pauli", 20, gs5a, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8)
        legs["C5"] = dict(auto_reset_leg(aenv, len(gs5a)), config=f"PauliGym 20q x {B}, difficulty 256 (targets regenerated on the device), compact_done + "
                                                                  "ptile_reset_tree_kernel + ptile_generate_kernel + ptile_step1c_kernel per pair")
        del aenv
        # ... and qg_vec_reset_done by itself where it is not a tree of row operations on a bit matrix: PauliGym 20q (config 5's env: a fresh target is
        # generated on the device), 1 % of the batch finished, eager calls (memset + compaction + two kernels), median of 12
        pg_n = 20
        pg_gs = gs5a
        penv = VecEnv("pauli", pg_n, pg_gs, B, add_perms=False, track_solution=False, difficulty=128)
        with torch.cuda.stream(stream):
            penv.reset(seed)
            pmask = (torch.rand(B, device=dev, generator=gen) < 0.01).to(torch.uint8)
            ptimes = []
            for i in range(12):
                penv.done.copy_(pmask)
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record(stream)
                penv.reset_done(seed + 100 + i)
                p1.record(stream)
                torch.cuda.synchronize()
                ptimes.append(p0.elapsed_time(p1) * 1e3)
        penv.sync()
        legs["pauli_reset_done"] = {"us_per_call": sorted(ptimes)[len(ptimes) // 2], "finished": float(pmask.float().mean()),
                                    "config": f"PauliGym {pg_n}q x {B} envs, difficulty 128, qg_vec_reset_done with 1 % of the batch finished, eager, median of 12"}
        del penv
        auto_reset = dict(legs["desynchronised"], synchronised=legs["synchronised"], reference_defaults=legs["desynchronised_reference_defaults"],
                          C2=legs["C2"], C2_x65536=legs["C2_x65536"], C2_reference_defaults=legs["C2_reference_defaults"], C5=legs["C5"],
                          pauli_reset_done=legs["pauli_reset_done"],
                          config=f"the headline workload with qg_vec_reset_done after every step (the pair reset_done + next step issued as qg_vec_reset_done_step: one launch), episodes of min(depth_slope * difficulty, max_depth) = {AT} steps, "
                                 f"a captured graph of {AT} x (step, reset_done) replayed 4 times; headline figures: episode ends spread evenly over time "
                                 "(parity of this schedule: tests/test_gpu_fullsize.py::test_auto_reset_with_desynchronised_episodes_at_full_size)")

    # ---- SURVEY 8(d) "report both modes": the observation handed to the learner after EVERY step of the headline workload.  The Gym adapter
    # returns the dense int8 [32, 32] matrix per env on every step() (adapters.py:50-54,62-72).  Four graphs of 128 steps each:
    #   packed        step + qg_vec_observe_packed (the bit-packed [B, 32] row words in env-major order: what the all-gather moves)
    #   dense         step + qg_vec_observe_dense  (8d's 1 184 B per env-step: a full 1 KiB rewrite per env)
    #   dense_tracked step on a handle with qg_vec_track_dense: the step kernel itself rewrites the <= 4 rows its gate changed in a RESIDENT
    #                 dense observation (same bytes in the tensor after every step, <= 128 of the 1 024 written)
    #   dense_kernel  qg_vec_observe_dense alone (launch period): the HBM-write roofline of the rewrite kernel on the bytes it writes
    obs_modes = None
    if not multi and B == ENVS_PER_GPU and not args.no_dense_obs:
        OT = 128
        D2 = 4 * n * n  # dense bytes per env
        oenv = VecEnv("clifford", n, gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
        tenv = VecEnv("clifford", n, gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
        obs_d = torch.empty((B, 2 * n, 2 * n), dtype=torch.int8, device=dev)
        obs_p = torch.empty((B, 2 * n), dtype=torch.int32, device=dev)

        def graph_of(body, replays=4):
            with torch.cuda.stream(stream):
                body()  # eager pass
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=stream):
                    body()
                torch.cuda.synchronize()
                gr.replay()
                torch.cuda.synchronize()
                o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                o0.record(stream)
                for _ in range(replays):
                    gr.replay()
                o1.record(stream)
            torch.cuda.synchronize()
            return o0.elapsed_time(o1) * 1e3 / (replays * OT)

        def steps_with(env_, after):
            def body():
                for t in range(OT):
                    env_.step(actions[t % RING])
                    after()
            return body

        with torch.cuda.stream(stream):
            oenv.reset(seed)
            tenv.reset(seed)
            tracked = tenv.track_dense()
        us_packed = graph_of(steps_with(oenv, lambda: oenv.observe_packed(out=obs_p)))
        us_dense = graph_of(steps_with(oenv, lambda: oenv.observe(out=obs_d)))
        us_kernel = graph_of(lambda: [oenv.observe(out=obs_d) for _ in range(OT)])
        us_tracked = graph_of(steps_with(tenv, lambda: None))
        # ... and with the reference's default options (add_inverts=True, track_solution=True; coins given): the two-lanes-per-env step rewrites
        # an env's whole observation when its coin inverts the matrix (half of the envs per step), the gate's rows otherwise
        denv2 = VecEnv("clifford", n, gateset, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=SCRAMBLE, max_depth=7 * OT)
        dcoins = torch.randint(0, 2, (RING, B), dtype=torch.uint8, device=dev, generator=gen)
        with torch.cuda.stream(stream):
            denv2.reset(seed)
            tracked2 = denv2.track_dense()

        def default_steps():
            for t in range(OT):
                denv2.step(actions[t % RING], dcoins[t % RING])

        us_tracked_default = graph_of(default_steps)
        denv2.sync()
        with torch.cuda.stream(stream):
            same2 = bool(torch.equal(tracked2, denv2.observe()))
        if not same2:
            raise SystemExit("bench.py: the tracked dense observation (reference-default options) differs from a full rewrite of the same state")
        del denv2, tracked2, dcoins
        oenv.sync()
        tenv.sync()
        with torch.cuda.stream(stream):
            oenv.observe_packed(out=obs_p)  # the packed words of the final state (the packed graph ran before the dense one)
            same = bool(torch.equal(tracked, tenv.observe())) and bool(torch.equal(obs_d, oenv.observe()))
        if not same:
            raise SystemExit("bench.py: the tracked dense observation differs from a full rewrite of the same state")
        # oracle check of every env: oenv and tenv took the same steps from the same reset (eager pass + 6 replays of each graph)
        obs_parity = None
        if rank == 0 and not args.no_parity:
            o_steps = 2 * 6 * OT  # oenv: two stepping graphs (eager pass + 5 replays each); tenv: one
            t_steps = 6 * OT
            sample = np.arange(B)  # every env
            acts_np = host_actions[:, sample].numpy()
            ov_o, _ = oracle_replay(gateset, seed, env_base + sample, acts_np, [t % RING for t in range(OT)] * (o_steps // OT))
            ov_t, _ = oracle_replay(gateset, seed, env_base + sample, acts_np, [t % RING for t in range(OT)] * (t_steps // OT))
            sidx = torch.as_tensor(sample, device=dev)
            ok_d = bool(np.array_equal(obs_d[sidx].cpu().numpy().reshape(len(sample), -1), ov_o.observe_dense()))
            ok_t = bool(np.array_equal(tracked[sidx].cpu().numpy().reshape(len(sample), -1), ov_t.observe_dense()))
            ok_p = bool(np.array_equal(obs_p[sidx].cpu().numpy().view(np.uint32), pack_rows_u32(ov_o.observe_dense().reshape(len(sample), 2 * n, 2 * n))))
            obs_parity = {"envs": int(len(sample)), "dense_rewrite": ok_d, "dense_tracked": ok_t, "packed": ok_p, "bit_exact": ok_d and ok_t and ok_p}
            if not obs_parity["bit_exact"]:
                raise SystemExit(f"bench.py: observation modes differ from the CPU oracle replay: {obs_parity}")

        def mode(us, bytes_per_env, what):
            gbs = bytes_per_env * B / (us * 1e-6) / 1e9
            return {"us_per_step": us, "value": B / (us * 1e-6), "unit": "env-steps/s", "bytes_per_env_step": bytes_per_env,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}, "what": what}

        obs_modes = {
            "packed": mode(us_packed, ALGO_BYTES_PER_STEP + 2 * 4 * 2 * n,
                           "step + qg_vec_observe_packed per step: 8d's 160 B + the env-major copy of the 32 row words (128 B read, 128 B written)"),
            "dense": mode(us_dense, ALGO_BYTES_PER_STEP + D2, "step + qg_vec_observe_dense per step: SURVEY 8d's 1 184 B per env-step (full 1 KiB int8 rewrite)"),
            "dense_tracked": {"us_per_step": us_tracked, "value": B / (us_tracked * 1e-6), "unit": "env-steps/s",
                              "bytes_moved_per_env_step": NEEDED_BYTES_PER_STEP + 64,
                              "frac_moved": (NEEDED_BYTES_PER_STEP + 64) * B / (us_tracked * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              "what": "qg_vec_track_dense: the step kernel rewrites the rows its gate changed in a resident dense observation; the tensor "
                                      "holds the same bytes as after the full rewrite.  No fraction on 8d's 1 184 B: this form does not move most of them; "
                                      "frac_moved is on the bytes it has to move"},
            "dense_tracked_reference_defaults": {
                "us_per_step": us_tracked_default, "value": B / (us_tracked_default * 1e-6), "unit": "env-steps/s",
                "what": "qg_vec_track_dense with add_inverts=True, track_solution=True (coins given): qm_inv2_kernel rewrites the whole env when its coin fires "
                        "(wave-contiguous 1 KiB stores), the gate's rows otherwise; equals a full rewrite of the final state (checked)"},
            "dense_kernel": {"kernel": "qg::qm_dense_stream_kernel<2>", "us_per_launch": us_kernel, "bytes_written_per_launch": D2 * B,
                             "roofline": {"bound": "hbm", "achieved": D2 * B / (us_kernel * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": D2 * B / (us_kernel * 1e-6) / 1e9 / HBM_PEAK_GBS},
                             "what": "launch period of qg_vec_observe_dense alone (hipGraph of 128 launches); written bytes only (it reads 128 B per env)"},
            "parity": obs_parity,
            "config": f"the headline workload, one observation after every step, hipGraphs of {OT} steps replayed 4 times; the average changed-row count of the "
                      "170-action gateset is 1.98 rows of 32 B per step",
        }
        del oenv, tenv, obs_d, obs_p, tracked

    # ---- SURVEY 8(d)'s other configurations: live launch period (hipGraph of 128 single-step launches) beside the committed rocprofv3 / PMC
    # figures of the same kernels (profiles/r05/traffic.json, tools/profile_bench.sh) -----
    configs = None
    if not multi and B == ENVS_PER_GPU and not args.no_configs:
        from util import line_gateset

        prof = profiled_configs()
        configs = {}

        def leg(name, venv, acts, coins=None):
            with torch.cuda.stream(stream):
                if coins is None:
                    us = graph_period(venv, acts)
                else:
                    venv.rollout(acts, coins=coins)  # builds the graph
                    tot = 0.0
                    for _ in range(3):
                        venv.reset(seed)  # a new episode: the solution log is empty again (not timed)
                        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        c0.record(stream)
                        venv.rollout(acts, coins=coins)
                        c1.record(stream)
                        torch.cuda.synchronize()
                        tot += c0.elapsed_time(c1)
                    venv.sync()
                    us = tot * 1e3 / (3 * acts.shape[0])
            p = prof.get(name) or {}
            st = p.get("rocprof_kernel_stats") or {}
            row = {"kernel": p.get("kernel"), "envs": venv.batch, "us_per_step": us, "value": venv.batch / (us * 1e-6), "unit": "env-steps/s",
                   "rocprof_avg_us": st.get("avg_us"), "pmc_bytes_per_env": p.get("bytes_per_env"), "needed_bytes_per_env": p.get("needed_bytes_per_env"),
                   "survey_8d_bytes_per_env": p.get("survey_8d_bytes_per_env"), "frac_moved": p.get("rocprof_frac_moved"),
                   "frac_survey_8d": p.get("rocprof_frac_algorithmic"), "source": "profiles/r05/traffic.json" if p else None}
            configs[name] = row

        gs2 = line_gateset("linear_function", 8)
        e2 = VecEnv("linear_function", 8, gs2, 8192, add_inverts=False, add_perms=False, track_solution=False, difficulty=64)
        with torch.cuda.stream(stream):
            e2.reset(0x5EED0002)
        leg("C2", e2, torch.randint(0, len(gs2), (RING, 8192), dtype=torch.int32, device=dev, generator=gen))
        del e2
        gs5 = line_gateset("pauli", 20)
        e5 = VecEnv("pauli", 20, gs5, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8)
        with torch.cuda.stream(stream):
            e5.reset(0x5EED0005)  # targets generated on the device: 1-7 rotations per env, tableau scrambled by 256 gates
        leg("C5", e5, torch.randint(0, len(gs5), (RING, B), dtype=torch.int32, device=dev, generator=gen))
        del e5
        e3d = VecEnv("clifford", n, gateset, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=SCRAMBLE)
        with torch.cuda.stream(stream):
            e3d.reset(seed)
        leg("C3d", e3d, torch.randint(0, A, (128, B), dtype=torch.int32, device=dev, generator=gen),
            coins=torch.randint(0, 2, (128, B), dtype=torch.uint8, device=dev, generator=gen))
        del e3d
        configs["note"] = ("C2 LinearFunctionGym 8q x 8 192, C5 PauliGym 20q x 65 536 (device-generated targets), C3d CliffordGym 16q x 65 536 with the "
                           "reference's default add_inverts=True / track_solution=True; us_per_step is live (HIP events around hipGraph replays of "
                           "single-step launches), the other columns are the committed rocprofv3 --kernel-trace --stats and --pmc passes of tools/run_config.py")

    # ---- the same step kernel at larger batches: where the launch boundary (1.6 us) stops dominating, and beyond the Infinity Cache -----
    large = None
    if not multi and B == ENVS_PER_GPU and not args.no_large_batch:
        sizes = []
        for LB in (1 << 18, 1 << 20, 1 << 22):
            big = VecEnv("clifford", n, gateset, LB, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
            bacts = torch.randint(0, A, (RING, LB), dtype=torch.int32, device=dev, generator=gen)
            with torch.cuda.stream(stream):
                big.reset(seed)
            lus = graph_period(big, bacts)
            lgb = ALGO_BYTES_PER_STEP * LB / (lus * 1e-6) / 1e9
            sizes.append({"envs": LB, "launch_us": lus, "achieved": lgb, "unit": "GB/s", "frac": lgb / HBM_PEAK_GBS,
                          "state_MiB": LB * 128 / 2**20, "rocprof": rocprof_kernel_avg_us(LB), "traffic": pmc_traffic(LB)})
            del big, bacts
        large = {"note": "same kernel and layout at 4x / 16x / 64x the batch: per-launch time is kernel time, not launch boundary; "
                         "2^22 envs = 512 MiB of state, beyond the 256 MiB Infinity Cache", "by_batch": sizes}

    # ---- with a policy in the loop: the reference's default network shape (rl/configs.py:531-607, bf16, random weights) forward +
    # categorical draw + env.step() + auto-reset per collection step, everything on the GPU; informational, not the headline metric -----
    collector = None
    if not multi and B == ENVS_PER_GPU and not args.no_collector:
        from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
        rows = []
        for CB in (1024, B):  # the reference's num_episodes (rl/configs.py:134) and the headline batch
            cenv = VecEnv("clifford", n, gateset, CB, add_inverts=False, add_perms=False, track_solution=False, difficulty=32)
            col = RolloutCollector(cenv, BasicPolicy(4 * n * n, A), dtype=torch.bfloat16, seed=1, store_obs="packed", use_graph=True)
            CT = 32
            col.collect(CT)  # eager pass + capture
            col.collect(CT)  # first replay (the graph's one-time upload: tens of ms now and then)
            torch.cuda.synchronize()
            per_replay = []
            for _ in range(5):  # each replay timed on its own: a hipGraph launch now and then stalls for tens of ms (its upload), the median does not see it
                t0 = time.perf_counter()
                ro = col.collect(CT)
                torch.cuda.synchronize()
                per_replay.append((time.perf_counter() - t0) / CT * 1e6)
            cus = float(np.median(per_replay))
            cenv.sync()
            rows.append({"envs": CB, "us_per_step": cus, "value": CB / (cus * 1e-6), "unit": "env-steps/s", "done_per_step": float(ro.dones.float().mean())})
            del col, cenv, ro
            torch.cuda.empty_cache()
        collector = {"policy": f"BasicPolicy {4 * n * n}-512-256-{{{A}, 1}} bf16, random weights; packed observation stored per step; one hipGraph per 32-step collection",
                     "clock": "host wall clock around each of 5 replays (device idle before and after), median", "by_batch": rows}

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1 and not args.force_multi:  # the CPU leg runs at N = 1 only
            cpu = cpu_baseline(gateset, seed)
        parity = None
        if snap is not None:
            parity = parity_replay(gateset, seed, env_base + ids, snap_actions, snap_trace, snap)
            if not parity["bit_exact"]:
                raise SystemExit(f"bench.py: GPU run differs from the CPU oracle replay: {parity}")
            if gshard is not None:
                parity["gathered_shard"] = gathered_parity(gateset, seed, env_base + ids, snap_actions, gshard["trace"], gshard)
                if not parity["gathered_shard"]["bit_exact"]:
                    raise SystemExit(f"bench.py: the all-gathered shard differs from the CPU oracle replay: {parity['gathered_shard']}")
                if "peer" in gshard:
                    pg = gshard["peer"]
                    parity["gathered_shard_of_last_rank"] = gathered_parity(gateset, seed, pg["global_ids"], pg["actions"], pg["trace"], pg)
                    if not parity["gathered_shard_of_last_rank"]["bit_exact"]:
                        raise SystemExit(f"bench.py: the last rank's part of the gathered shard differs from the CPU oracle replay: {parity['gathered_shard_of_last_rank']}")
        graphs = [c for kind, c in timed_launches if kind == "graph"]
        eager = sum(c for kind, c in timed_launches if kind == "eager")
        launch_desc = ("one step kernel per env.step(); the %d timed steps = %s%s" % (
            K, " + ".join(f"{graphs.count(c)} x hipGraph of {c} launches" for c in sorted(set(graphs), reverse=True)) or "no graph",
            f" + {eager} eager launch(es)" if eager else ""))
        total_steps = B * K * n_gpus
        # the committed profile is evidence, not the measurement: it must agree with what this run measured (the rocprofv3 average
        # brackets between the steady-state launch period and the eager launch's event time), or it is stale -- said, not fatal
        rocprof_vs_live = None
        if rocprof:
            lo, hi = 0.75 * min(graph_period_us, timed_region_us), 1.35 * max(eager_event_us, timed_region_us)
            rocprof_vs_live = dict(rocprof, frac=algo_bytes / (rocprof["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   agrees_with_live=bool(lo <= rocprof["avg_us"] <= hi), live_bracket_us=[lo, hi])
            if not rocprof_vs_live["agrees_with_live"]:
                print(f"bench.py: WARNING: {rocprof['source']} says {rocprof['avg_us']:.2f} us per launch, this run measured {graph_period_us:.2f} (graph period) .. "
                      f"{eager_event_us:.2f} (eager launch): the committed profile is stale -- re-run tools/profile_bench.sh", file=sys.stderr)
        shared = bool(args.ranks_share_gpu0 and world > 1)
        out = {
            "metric": (f"env-steps/sec ({world} ranks time-sharing ONE GPU: a functional record of the multi-rank path, not a scaling number), "
                       f"CliffordGym 16q x {B} envs/rank; bit-exact vs CPU") if shared else
                      f"env-steps/sec (whole node), CliffordGym 16q x {B} envs/GPU; bit-exact vs CPU",
            "value": total_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": 1 if shared else n_gpus,  # distinct devices in use
            "ranks": world,
            "physical_gpus": 1 if shared else n_gpus,
            "ranks_share_gpu0": shared,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "CliffordGym 16 qubits, line-16 bidirectional, 170 actions (H,S,Sdg,SX,SXdg,CX,CZ,SWAP), "
                            f"{B} envs per GPU, start = identity + {SCRAMBLE} random gates, uniform random actions "
                            f"(ring of {RING} pre-sampled action buffers, as a policy rewriting one resident buffer per step), "
                            "add_inverts=False, add_perms=False, track_solution=False, default weights, free-running",
                "envs_per_gpu": B,
                "total_envs": B * n_gpus,
                "ranks_seen": dist.get_world_size() if dist is not None else 1,
                "env_ids_of_rank0": [env_base, env_base + B],
                "partition": f"rank r owns envs [r * {B}, (r + 1) * {B}) of one batch of {total_envs}; seeds and actions are functions of the global env id",
                "launch": launch_desc,
                "collective": None if not multi or args.no_gather else {
                    "handover": args.handover,
                    "what": "qg_vec_push_learner_shard + qg_comm_p2p_wait + qg_comm_p2p_release (include/qgym.h) on the step stream: the packed shard copied "
                            "into every rank's hipIpc window, per-source arrival flags; handles exchanged over the control plane (gloo); no RCCL" if args.handover == "direct" else
                            "qg_comm_gather_submit (include/qgym.h): one flat shard per rank (bit-packed observation + f32 rewards + is_final / success flags) "
                            "packed by libqgym and moved by ncclAllGather of librccl.so.1 called from libqgym, side stream, double buffered, "
                            "overlapped with the following steps; torch.distributed (gloo) carries only the id, the barriers and the timing reduction",
                    "every_steps": gather_every,
                    "collectives_in_timed_region": gathers_timed,
                },
            },
            "roofline": {
                "bound": "hbm",
                "kernel": KERNEL,
                "kernel_resources": "256 threads/block, 1 wave/SIMD at 65 536 envs; no LDS; thread per env",
                # achieved / frac: SURVEY 8(d)'s algorithmic bytes per launch over the step kernel's DURATION ON THE DEVICE CLOCK measured in this run
                # (qg_vec_set_kernel_clock: first wave entry -> last wave exit, stamped by the waves; mean of 4 x 256 launches).  The launch period
                # (kernel + launch boundary: HIP events on the launch stream around the K timed steps -- what `value` counts) is beside it, and the
                # committed rocprofv3 --kernel-trace --stats average of the same command (tools/profile_bench.sh -> profiles/)
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "clock": "kernel device clock (qg_vec_set_kernel_clock): last wave exit - first wave entry on the 100 MHz constant-rate counter, "
                         "every wave waiting for its own loads and stores before its exit stamp; mean over 4 replays of a 256-launch hipGraph in this run",
                "kernel_us_device_clock": {"mean": device_clock_us, "median": float(np.median(kdurs)), "min": float(kdurs.min()), "max": float(kdurs.max()),
                                           "p10": float(np.percentile(kdurs, 10)), "p90": float(np.percentile(kdurs, 90)), "launches": int(kdurs.size)},
                "launch_period_us": timed_region_us,
                "frac_launch_period": algo_bytes / (timed_region_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic["bytes_per_launch"] if traffic else None,
                "traffic_detail": traffic,
                "algorithmic_bytes_per_launch": algo_bytes,
                "bytes_needed_per_launch": needed_bytes,
                "bytes_per_env": {"survey_8d": ALGO_BYTES_PER_STEP, "needed": NEEDED_BYTES_PER_STEP,
                                  "pmc": traffic["bytes_per_launch"] / B if traffic else None},
                "traffic_over_needed": traffic["bytes_per_launch"] / needed_bytes if traffic else None,
                "kernel_us_timed_region": timed_region_us,
                "kernel_us_graph_period": graph_period_us,
                "kernel_us_eager_event": eager_event_us,
                "rocprof_committed": rocprof_vs_live,
                "frac_by_clock": {
                    "device_clock": achieved / HBM_PEAK_GBS,
                    "timed_region": algo_bytes / (timed_region_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "graph_period": algo_bytes / (graph_period_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "eager_event": algo_bytes / (eager_event_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "rocprof_committed_avg": algo_bytes / (rocprof["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS if rocprof else None,
                },
            },
            "cpu_baseline": cpu,
            "parity": parity,
            "fused_rollout": fused,
            "default_config": default_cfg,
            "auto_reset": auto_reset,
            "observation_modes": obs_modes,
            "configs": configs,
            "large_batch": large,
            "policy_in_loop": collector,
        }

        def r2(x):
            return None if x is None else round(float(x), 2)

        # the figures of the legs above once more, compact and LAST in the line: whoever keeps only the line's tail still has them
        out["summary"] = {
            "us_per_step": r2(elapsed * 1e6 / K), "launch_period_us": r2(timed_region_us), "kernel_us_device_clock": r2(device_clock_us),
            "roofline_frac": round(achieved / HBM_PEAK_GBS, 3), "parity_envs": parity["envs"] if parity else None,
            "cpu_env_steps_per_s": cpu["value"] if cpu else None, "cpu_cores": cpu["cores"] if cpu else None,
            "default_config_us": r2(default_cfg["us_per_step"]) if default_cfg else None,
            "auto_reset_us_per_pair": None if not auto_reset else {
                "C3": r2(auto_reset["us_per_step"]), "C3_synchronised": r2(auto_reset["synchronised"]["us_per_step"]),
                "C3_reference_defaults": r2(auto_reset["reference_defaults"]["us_per_step"]), "C2": r2(auto_reset["C2"]["us_per_step"]),
                "C2_x65536": r2(auto_reset["C2_x65536"]["us_per_step"]), "C2_reference_defaults": r2(auto_reset["C2_reference_defaults"]["us_per_step"]),
                "C5": r2(auto_reset["C5"]["us_per_step"]), "pauli_reset_done_1pct_eager": r2(auto_reset["pauli_reset_done"]["us_per_call"])},
            "observation_us_per_step": None if not obs_modes else {k: r2(obs_modes[k]["us_per_step"]) for k in ("packed", "dense", "dense_tracked", "dense_tracked_reference_defaults")},
            "dense_rewrite_kernel_frac": round(obs_modes["dense_kernel"]["roofline"]["frac"], 3) if obs_modes else None,
            "configs_us_per_step": None if not configs else {k: r2(configs[k]["us_per_step"]) for k in ("C2", "C5", "C3d")},
            "large_batch": None if not large else {str(b["envs"]): {"us": r2(b["launch_us"]), "frac": round(b["frac"], 3)} for b in large["by_batch"]},
            "fused_rollout_env_steps_per_s": fused["value"] if fused else None,
            "policy_in_loop_us_per_step": None if not collector else {str(b["envs"]): r2(b["us_per_step"]) for b in collector["by_batch"]},
        }
    else:
        out = None

    # ---- the optional collective cadences (N > 1), last and under a watchdog: whatever happens in there, rank 0's line goes out -----
    printed = threading.Event()

    def emit(cadence):
        if out is None or printed.is_set():
            return
        printed.set()
        if out["config"]["collective"] is not None and cadence:
            out["config"]["collective"].update(cadence)
        print(json.dumps(out), file=json_out, flush=True)

    def out_of_time():
        emit({"cadence_error": f"the collective-cadence legs did not finish within {CADENCE_BUDGET_S} s (a rank or a device-side wait is stuck); "
                               "the line above them is complete"})
        print(f"bench.py: rank {rank}: cadence legs exceeded {CADENCE_BUDGET_S} s, leaving", file=sys.stderr)
        sys.stderr.flush()
        os._exit(4)

    cadence = None
    if multi:
        watchdog = threading.Timer(CADENCE_BUDGET_S, out_of_time)
        watchdog.daemon = True
        watchdog.start()
        try:
            cadence = collective_cadences()
        except Exception as exc:  # noqa: BLE001 -- optional legs: reported in the line
            cadence = {"cadence_error": f"{type(exc).__name__}: {exc}"}
    emit(cadence)
    if dist is not None:
        try:
            torch.cuda.synchronize()
            dist.barrier()  # nobody unmaps a window or leaves the communicator while a peer may still use it
        except Exception as exc:  # noqa: BLE001 -- a peer is gone: nothing left to protect
            print(f"bench.py: rank {rank}: final barrier failed: {exc}", file=sys.stderr)
            os._exit(5)
        watchdog.cancel()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
