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

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): every rank owns 65 536 envs
(weak scaling); env.step needs no collective.  The exchange BASELINE.json's north_star names --
the observation handed back to the learner -- is an all-gather of the bit-packed observation
(8 MiB per rank) on a side stream, double buffered and overlapped with the following steps, once
per rollout segment (--gather-every, default 256 steps; 1 = after every step, which is link-bound
at >= 55 us per step against a 4 us step).

Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

NUM_QUBITS = 16
ENVS_PER_GPU = 65536
SCRAMBLE = 256
CHUNK = 256  # steps per hipGraph replay (single-GPU path)
RING = 16    # pre-sampled action buffers the steps cycle through (a policy rewrites ONE buffer per step)
ALGO_BYTES_PER_STEP = 160  # 128 B state read + 16 B touched rows + 16 B scalars
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def build_gateset():
    from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map, line_edges

    kinds = ["H", "S", "Sdg", "SX", "SXdg", "CX", "CZ", "SWAP"]
    return gateset_from_coupling_map(line_edges(NUM_QUBITS, True), None, kinds)


def cpu_baseline(gateset, seed: int, budget_s: float = 12.0):
    """Time the CPU oracle (a C port of the reference's scalar Rust path, one env object per env,
    OpenMP over envs like twisterl's rayon-over-clones) on this box's host cores."""
    from oracle import OracleEnv, OracleVec

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    B = 16384
    A = len(gateset)
    proto = OracleEnv("clifford", NUM_QUBITS, gateset, add_inverts=0, add_perms=0, track_solution=0, difficulty=SCRAMBLE)
    ov = OracleVec(proto, B)
    rng = np.random.default_rng(seed)
    ov.reset_with(rng.integers(0, A, size=(SCRAMBLE, B)))
    acts = rng.integers(0, A, size=(32, B)).astype(np.int32)
    for t in range(8):  # warm-up: thread pool, first-touch, allocator
        ov.step_only(acts[t], threads=min(avail, 8))
    # the box may expose more hardware threads than it lets one container run: pick the thread
    # count that actually delivers the most steps/s and report that count as `cores`
    best = (0.0, 1)
    cand = sorted({c for c in (1, 2, 4, 8, 16, 32, 64, 128, 256) if c <= avail} | {min(avail, 256)})
    for c in cand:
        ov.step_only(acts[0], threads=c)
        t0 = time.perf_counter()
        for t in range(4):
            ov.step_only(acts[t], threads=c)
        rate = 4 * B / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, c)
    cores = best[1]
    per_step = B / best[0]
    n_steps = int(max(8, min(2_000_000, budget_s / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for t in range(n_steps):
        ov.step_only(acts[t % 32], threads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": B * n_steps / dt,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"CliffordGym 16q, {B} envs x {n_steps} steps ({dt:.1f} s), C port of the reference scalar path "
                  f"(byte-per-entry state, per-env objects), OpenMP over envs on {cores} threads",
    }


def parity_replay(gateset, seed, ids, ring_actions, trace, snap):
    """Replay the run the GPU just did -- same seed, same scramble draws, same action buffers in the
    same order -- on the CPU oracle for the sampled envs, and compare everything env.step() produces
    (reward bits, success, is_final, depth, dense observation) after the last timed step."""
    from oracle import OracleEnv, OracleVec
    from util import f32_bits, rng_actions

    proto = OracleEnv("clifford", NUM_QUBITS, gateset, add_inverts=0, add_perms=0, track_solution=0, difficulty=SCRAMBLE)
    ov = OracleVec(proto, len(ids))
    ov.reset_with(rng_actions(seed, ids, SCRAMBLE, len(gateset)))
    r = s = f = d = None
    for ring_idx in trace:
        r, s, f, d = ov.step(ring_actions[ring_idx])
    ok = {
        "reward_bits": bool(np.array_equal(f32_bits(snap["reward"]), f32_bits(r))),
        "success": bool(np.array_equal(snap["success"], s)),
        "is_final": bool(np.array_equal(snap["done"], f)),
        "depth": bool(np.array_equal(snap["depth"], d)),
        "observation": bool(np.array_equal(snap["obs"].reshape(len(ids), -1), ov.observe_dense())),
    }
    import hashlib

    def digest(obs, reward, success, depth):  # SHA-256 over the final (state, reward bits, success, depth) streams
        h = hashlib.sha256()
        for arr in (np.asarray(obs, dtype=np.uint8), f32_bits(reward).astype(np.uint32), np.asarray(success, dtype=np.uint8), np.asarray(depth, dtype=np.int32)):
            h.update(np.ascontiguousarray(arr).tobytes())
        return h.hexdigest()

    sha_gpu = digest(snap["obs"].reshape(len(ids), -1), snap["reward"], snap["success"], snap["depth"])
    sha_cpu = digest(ov.observe_dense(), r, s, d)
    ok["sha256"] = sha_gpu == sha_cpu
    return {"envs": int(len(ids)), "steps_replayed": len(trace), "checked": sorted(ok), "bit_exact": all(ok.values()),
            "mismatch": [k for k, v in ok.items() if not v], "sha256_hip": sha_gpu, "sha256_oracle": sha_cpu}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the CPU-oracle replay of the timed run")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the 2^20-env leg (profiling runs: keeps the kernel statistics to one batch size)")
    ap.add_argument("--no-gather", action="store_true", help="N>1 diagnostics: skip the per-step all-gather")
    ap.add_argument("--gather-every", type=int, default=CHUNK,
                    help="N>1: all-gather the packed observation every this many steps (1 = after every step)")
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU,
                    help="diagnostics: envs per GPU (the metric is quoted at the default 65 536; profiles/ uses 2^20 to show where the "
                         "launch boundary stops mattering)")
    ap.add_argument("--force-multi", action="store_true",
                    help="diagnostics: run the multi-GPU code path (RCCL init, side-stream all-gather) even with one rank")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print to the C-level stdout (RCCL's version
    # banner) are sent to stderr, our line goes to a private duplicate of the original descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    multi = world > 1 or args.force_multi
    if multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    n_gpus = world
    dev = torch.device("cuda", torch.cuda.current_device())

    from qiskit_gym_amd.vec import VecEnv

    n, gateset = build_gateset()
    A = len(gateset)
    B = args.envs
    seed = 0x5EED0003 + rank
    env = VecEnv("clifford", n, gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
    stream = torch.cuda.Stream(device=dev)
    K, W = args.steps, args.warmup

    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    actions = torch.randint(0, A, (RING, B), dtype=torch.int32, device=dev, generator=gen)

    ring_trace = []  # which action buffer every step since the last reset used (for the oracle replay)

    def run_steps_single(nsteps: int):
        """nsteps env.step() launches: whole chunks replay a cached hipGraph of CHUNK launches."""
        done = 0
        while nsteps - done >= CHUNK:
            env.rollout_ring(actions, CHUNK)
            ring_trace.extend(i % RING for i in range(CHUNK))
            done += CHUNK
        for t in range(nsteps - done):
            env.step(actions[t % RING])
            ring_trace.append(t % RING)

    # ---- multi-GPU: step + all-gather of the packed observation, double buffered -------------
    if multi:
        from qiskit_gym_amd.distributed import OverlappedGather

        # double-buffered, host-mediated hand-over to a side stream (see OverlappedGather: a stream-to-stream
        # event wait would slow every later graph replay on the step stream by ~40 %)
        # one flat int32 shard per rank carries everything SURVEY 8e lists for the learner: the packed observation [B, 32], the f32
        # rewards [B] and is_final / success [B] bytes each -- one collective instead of four
        from qiskit_gym_amd.distributed import fill_learner_shard, learner_shard_words

        gatherer = OverlappedGather((learner_shard_words(B, 32),), torch.int32, dev)

        def fill_shard(buf):
            fill_learner_shard(buf, B, 32, lambda view: env.observe_packed(out=view), env.reward, env.done, env.success)

        def snapshot_and_gather():
            gatherer.submit(fill_shard)

        flush_gathers = gatherer.flush

        def run_steps_multi(nsteps: int):
            """Each rank steps its own shard (no collective inside step).  Every `gather_every` steps the
            bit-packed observation is snapshotted and all-gathered on the side stream, double buffered,
            overlapping the steps that follow."""
            G = args.gather_every
            done = 0
            while done < nsteps:
                n = min(G, nsteps - done)
                if n >= 8:
                    env.rollout_ring(actions, n)  # cached hipGraph of n single-step launches
                    ring_trace.extend(i % RING for i in range(n))
                else:
                    for t in range(n):
                        env.step(actions[(done + t) % RING])
                        ring_trace.append((done + t) % RING)
                done += n
                if not args.no_gather and n == G:
                    snapshot_and_gather()
            flush_gathers()

        run_steps = run_steps_multi
    else:
        run_steps = run_steps_single

    with torch.cuda.stream(stream):
        env.reset(seed)
        run_steps(CHUNK)  # builds and caches the rollout graph (setup, not a step)
        if multi and not args.no_gather:  # communicator set-up and first-use kernel loads of RCCL (setup, not a step)
            snapshot_and_gather()
            snapshot_and_gather()
            flush_gathers()
        env.reset(seed)
        ring_trace.clear()
        run_steps(W)  # untimed warmup steps
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        ev0.record(stream)
        run_steps(K)
        ev1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    env.sync()  # raises if any env faulted
    stream_ms = ev0.elapsed_time(ev1)

    # ---- snapshot of a sample of envs right after the timed steps, for the oracle replay -------
    snap = None
    if rank == 0 and not args.no_parity:
        ids = np.arange(0, B, 64)
        idx = torch.as_tensor(ids, device=dev)
        with torch.cuda.stream(stream):
            snap = {
                "obs": env.observe()[idx].cpu().numpy(),
                "reward": env.reward[idx].cpu().numpy(),
                "success": env.success[idx].cpu().numpy(),
                "done": env.done[idx].cpu().numpy(),
                "depth": env.depth[idx].cpu().numpy(),
            }
        snap_trace = list(ring_trace)
        snap_actions = actions[:, idx].cpu().numpy()

    # ---- roofline leg: duration of the step kernel itself, HIP events around single launches --
    reps = 200
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    with torch.cuda.stream(stream):
        for i in range(reps):
            starts[i].record(stream)
            env.step(actions[i % RING])
            stops[i].record(stream)
    torch.cuda.synchronize()
    per_launch_us = sorted(s.elapsed_time(e) * 1e3 for s, e in zip(starts, stops))
    single_launch_us = float(np.median(per_launch_us))
    b2b_us = stream_ms * 1e3 / K  # back-to-back launches incl. the inter-kernel boundary
    kernel_us = min(single_launch_us, b2b_us)
    algo_bytes = ALGO_BYTES_PER_STEP * B
    achieved = algo_bytes / (kernel_us * 1e-6) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and B == ENVS_PER_GPU:  # the PMC passes were collected at the metric's batch size
        try:
            traffic = json.load(open(tpath)).get("clifford_step_bytes_per_launch")
        except Exception:
            traffic = None

    # ---- fused rollout (state in registers across steps), reported beside the headline --------
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

    # ---- the same step kernel at 2^18 and 2^20 envs: where the launch boundary (1.6 us) stops dominating -----
    large = None
    if not multi and B == ENVS_PER_GPU and not args.no_large_batch:
        sizes = []
        for LB in (1 << 18, 1 << 20):
            big = VecEnv("clifford", n, gateset, LB, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
            bacts = torch.randint(0, A, (RING, LB), dtype=torch.int32, device=dev, generator=gen)
            with torch.cuda.stream(stream):
                big.reset(seed)
                big.rollout_ring(bacts, CHUNK)
                torch.cuda.synchronize()
                b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                b0.record(stream)
                for _ in range(2):
                    big.rollout_ring(bacts, CHUNK)
                b1.record(stream)
            torch.cuda.synchronize()
            big.sync()
            lus = b0.elapsed_time(b1) * 1e3 / (2 * CHUNK)
            lgb = ALGO_BYTES_PER_STEP * LB / (lus * 1e-6) / 1e9
            sizes.append({"envs": LB, "launch_us": lus, "achieved": lgb, "unit": "GB/s", "frac": lgb / HBM_PEAK_GBS})
            del big, bacts
        large = dict(sizes[-1], note="same kernel and layout, 16x the batch: per-launch time is kernel time, not launch boundary", by_batch=sizes)

    out = None
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N = 1 only
            cpu = cpu_baseline(gateset, seed)
        parity = None
        if snap is not None:
            parity = parity_replay(gateset, seed, ids, snap_actions, snap_trace, snap)
            if not parity["bit_exact"]:
                raise SystemExit(f"bench.py: GPU run differs from the CPU oracle replay: {parity}")
        total_steps = B * K * n_gpus
        out = {
            "metric": f"env-steps/sec (whole node), CliffordGym 16q x {B} envs/GPU; bit-exact vs CPU",
            "value": total_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": n_gpus,
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
                "launch": "one step kernel per env.step(); chunks of %d launches replayed from a hipGraph" % CHUNK
                if not multi else "one step kernel per env.step() (hipGraph chunks); no collective inside step",
                "collective": None if not multi or args.no_gather else
                f"RCCL all_gather_into_tensor of one shard per rank (bit-packed observation 8 MiB + rewards + is_final / success flags) every {args.gather_every} steps, side stream, overlapped",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "qg::qm_step1_kernel<16, true, false>",
                "kernel_resources": "256 threads/block, 1 wave/SIMD at 65 536 envs; no LDS; thread per env",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes,
                "launch_us": kernel_us,
                "launch_us_single_event_pair": single_launch_us,
                "launch_us_back_to_back": b2b_us,
            },
            "cpu_baseline": cpu,
            "parity": parity,
            "fused_rollout": fused,
            "large_batch": large,
        }
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
