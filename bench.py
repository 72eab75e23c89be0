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

import bench_common  # noqa: F401  (first: it sets the OpenMP / HSA environment before anything can load those runtimes)
from bench_common import *  # noqa: F401,F403,E402  constants, cpu_baseline, the oracle replay, the readers of the committed profiles
from bench_common import (ALGO_BYTES_PER_STEP, CADENCE_BUDGET_S, CHUNK, CONTROL_TIMEOUT_S, ENVS_PER_GPU, HBM_PEAK_GBS, KERNEL, NEEDED_BYTES_PER_STEP, NUM_QUBITS,  # noqa: E402
                          RING, ROOT, SCRAMBLE, build_gateset, cpu_baseline, gathered_parity, global_actions, parity_replay, pmc_traffic, rocprof_kernel_avg_us)

import argparse  # noqa: E402
import json  # noqa: E402
import os  # noqa: E402
import socket  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402
import threading  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


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

    # ---- the legs beside the headline (bench_legs.py; N = 1 at the metric's batch size) --------
    import bench_legs
    from types import SimpleNamespace

    ctx = SimpleNamespace(multi=multi, B=B, args=args, VecEnv=VecEnv, n=n, gateset=gateset, A=A, dev=dev, gen=gen, stream=stream, seed=seed, actions=actions,
                          host_actions=host_actions, env_base=env_base, rank=rank, env=env)
    fused = bench_legs.fused_rollout(ctx)
    default_cfg = bench_legs.default_config(ctx)
    auto_reset = bench_legs.auto_reset(ctx)
    obs_modes = bench_legs.observation_modes(ctx)
    configs = bench_legs.other_configs(ctx)
    large = bench_legs.large_batch(ctx)
    collector = bench_legs.policy_in_loop(ctx)

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
                "C5": r2(auto_reset["C5"]["us_per_step"]), "clifford24": r2(auto_reset["clifford24"]["us_per_step"]), "pauli_reset_done_1pct_eager": r2(auto_reset["pauli_reset_done"]["us_per_call"])},
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
