"""One authoritative kernel clock: the duration of the dominant kernel of every measured configuration as the kernel's own waves see it
(qg_vec_set_kernel_clock: first wave entry -> last wave exit on the device's constant-rate counter, every wave waiting for its own loads and
stores before its exit stamp), beside the two clocks that bracket it from outside:

  * the launch period (HIP events on the launch stream around hipGraph replays: kernel + launch boundary), with and without the stamps;
  * the committed rocprofv3 --kernel-trace --stats average of the same kernel (profiles/<round>/*_kernel_stats.csv), when present.

  python tools/kernel_device_clock.py [--out profiles/r05/kernel_device_clock] [--replays 8]

Writes <out>.txt (the table) and <out>.json.  No profiler involved: run it plainly.
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

HBM = 8000.0  # GB/s
T = 128


def rocprof_avg(profile_dir, fname, needle):
    try:
        calls, tot = 0, 0.0
        with open(os.path.join(profile_dir, fname)) as f:
            for row in csv.DictReader(f):
                if needle in row["Name"]:
                    calls += int(row["Calls"])
                    tot += float(row["AverageNs"]) * int(row["Calls"])
        return tot / calls / 1e3 if calls else None
    except OSError:
        return None


def measure(name, env, body, stream, replays, bytes_8d, kernel, prof, before=None):
    """body(): enqueue T launches of the kernel on the current stream; before(): untimed, ahead of every run of body / group of replays
    (a reset that empties the solution log).  Returns the row."""
    before = before or (lambda: None)

    def graph_of(stamped=False):
        """Eager pass, then the capture; `stamped`: the captured launches (and only they) carry kernel-clock slots 0 .. T-1."""
        with torch.cuda.stream(stream):
            before()
            body()
            torch.cuda.synchronize()
            slots = env.kernel_clock(T) if stamped else None  # (armed here: the eager pass and before() take no slot)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                body()
            torch.cuda.synchronize()
            before()
            g.replay()
            torch.cuda.synchronize()
        return (g, slots) if stamped else g

    def period(g, n=4):
        with torch.cuda.stream(stream):
            before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(n):
                g.replay()
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (n * T)

    g_plain = graph_of()
    period_plain = min(period(g_plain) for _ in range(3))
    del g_plain
    g, slots = graph_of(stamped=True)  # every captured launch carries its own slot
    period_stamped = min(period(g) for _ in range(3))
    durs = []
    for _ in range(replays):
        slots.zero_()
        with torch.cuda.stream(stream):
            before()
            torch.cuda.synchronize()
            g.replay()
        torch.cuda.synchronize()
        d = env.kernel_durations_us(slots)
        assert len(d) == T, (name, len(d))
        durs.append(d)
    env.kernel_clock(0)
    del g
    env.sync()
    durs = np.concatenate(durs)
    B = env.batch
    row = {"config": name, "kernel": kernel, "envs": B, "launches_stamped": int(durs.size),
           "device_clock_us": {"mean": float(durs.mean()), "median": float(np.median(durs)), "min": float(durs.min()), "max": float(durs.max()),
                               "p10": float(np.percentile(durs, 10)), "p90": float(np.percentile(durs, 90))},
           "launch_period_us": period_plain, "launch_period_with_stamps_us": period_stamped,
           "launch_boundary_us": period_plain - float(durs.mean()),
           "rocprof_committed_avg_us": prof, "bytes_8d_per_env": bytes_8d,
           "GBs_of_8d_bytes_device_clock": bytes_8d * B / (float(durs.mean()) * 1e-6) / 1e9 if bytes_8d else None,
           "frac_of_8TBs_device_clock": bytes_8d * B / (float(durs.mean()) * 1e-6) / 1e9 / HBM if bytes_8d else None,
           "frac_of_8TBs_launch_period": bytes_8d * B / (period_plain * 1e-6) / 1e9 / HBM if bytes_8d else None}
    if row["frac_of_8TBs_device_clock"] and row["frac_of_8TBs_device_clock"] > 1.0:
        # more 8d bytes per second than HBM can deliver: the launch's working set is absorbed by the 256 MiB Infinity Cache (or 8d's count exceeds what
        # the kernel moves) -- no fraction of the HBM peak is quoted for such a row
        row["frac_of_8TBs_device_clock"] = None
        row["cache_resident"] = True
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05", "kernel_device_clock"))
    ap.add_argument("--replays", type=int, default=8)
    ap.add_argument("--profile-dir", default=os.path.join(ROOT, "profiles", "r05"))
    ap.add_argument("--only", default=None, help="comma-separated subset of configuration names")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    plain = dict(add_inverts=False, add_perms=False, track_solution=False)
    rows = []
    want = set(args.only.split(",")) if args.only else None

    def ring(env, A, n=16):
        return torch.randint(0, A, (n, env.batch), dtype=torch.int32, device=dev, generator=gen)

    def steps(env, acts, coins=None):
        def body():
            for t in range(T):
                env.step(acts[t % acts.shape[0]], None if coins is None else coins[t % coins.shape[0]])
        return body

    def add(name, make):
        if want and name not in want:
            return
        rows.append(make())
        r = rows[-1]
        print(f"{name}: device clock {r['device_clock_us']['mean']:.2f} us, launch period {r['launch_period_us']:.2f} us", file=sys.stderr, flush=True)

    gs3 = line_gateset("clifford", 16)
    for B, fname in ((65536, "bench_kernel_stats.csv"), (1 << 20, "bench_1048576_kernel_stats.csv"), (1 << 22, "bench_4194304_kernel_stats.csv")):
        def mk(B=B, fname=fname):
            env = VecEnv("clifford", 16, gs3, B, difficulty=256, **plain)
            with torch.cuda.stream(stream):
                env.reset(0x5EED0003)
            return measure("C3" if B == 65536 else f"C3_{B}", env, steps(env, ring(env, len(gs3))), stream, args.replays, 160,
                           "qm_step1_kernel<16, true, false, false, false>", rocprof_avg(args.profile_dir, fname, "qm_step1_kernel<16, true, false"))
        add("C3" if B == 65536 else f"C3_{B}", mk)

    def mk_c2():
        gs = line_gateset("linear_function", 8)
        env = VecEnv("linear_function", 8, gs, 8192, difficulty=64, **plain)
        with torch.cuda.stream(stream):
            env.reset(0x5EED0002)
        return measure("C2", env, steps(env, ring(env, len(gs))), stream, args.replays, 32, "word_step_kernel<false>",
                       rocprof_avg(args.profile_dir, "C2_kernel_stats.csv", "word_step_kernel<false>"))
    add("C2", mk_c2)

    def mk_c3d():
        env = VecEnv("clifford", 16, gs3, 65536, difficulty=256, add_inverts=True, add_perms=False, track_solution=True, max_depth=4 * T)
        with torch.cuda.stream(stream):
            env.reset(0x5EED0003)
        coins = torch.randint(0, 2, (16, 65536), dtype=torch.uint8, device=dev, generator=gen)
        return measure("C3d", env, steps(env, ring(env, len(gs3)), coins), stream, args.replays, 160, "qm_inv2_kernel<16, true, false, false>",
                       rocprof_avg(args.profile_dir, "C3d_kernel_stats.csv", "qm_inv2_kernel<16, true"), before=lambda: env.reset(0x5EED0003))
    add("C3d", mk_c3d)

    for B in (65536, 1 << 20, 1 << 22):
        def mk_c5(B=B):
            gs = line_gateset("pauli", 20)
            env = VecEnv("pauli", 20, gs, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8)
            with torch.cuda.stream(stream):
                env.reset(0x5EED0005)
            nm = "C5" if B == 65536 else f"C5_{B}"
            return measure(nm, env, steps(env, ring(env, len(gs))), stream, args.replays, 494, "ptile_step1c_kernel<20, 8, false>",
                           rocprof_avg(args.profile_dir, "C5_kernel_stats.csv" if B == 65536 else f"C5_{B}_kernel_stats.csv", "ptile_step1c_kernel<20, 8"))
        add("C5" if B == 65536 else f"C5_{B}", mk_c5)

    def mk_dense():
        env = VecEnv("clifford", 16, gs3, 65536, difficulty=256, **plain)
        out = torch.empty((65536, 32, 32), dtype=torch.int8, device=dev)
        with torch.cuda.stream(stream):
            env.reset(0x5EED0003)

        def body():
            for _ in range(T):
                env.observe(out=out)
        return measure("dense", env, body, stream, args.replays, 1152, "qm_dense_stream_kernel<2>",
                       rocprof_avg(args.profile_dir, "dense_kernel_stats.csv", "qm_dense_stream_kernel<2>"))
    add("dense", mk_dense)

    def mk_tracked():
        env = VecEnv("clifford", 16, gs3, 65536, difficulty=256, **plain)
        with torch.cuda.stream(stream):
            env.reset(0x5EED0003)
            keep = env.track_dense()  # noqa: F841
        return measure("tracked", env, steps(env, ring(env, len(gs3))), stream, args.replays, 154 + 64, "qm_step1_kernel<16, true, false, false, true>",
                       rocprof_avg(args.profile_dir, "tracked_kernel_stats.csv", "qm_step1_kernel<16, true, false, false, true>"))
    add("tracked", mk_tracked)

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    rate = VecEnv("linear_function", 8, line_gateset("linear_function", 8), 64, **plain)._L.qg_kernel_clock_rate_khz(0)
    doc = {"clock": f"qg_vec_set_kernel_clock: s_memrealtime ({rate} kHz => {1e6 / rate:.0f} ns per tick), min over the waves' entry stamps to max over their "
                    "exit stamps (each wave after s_waitcnt vmcnt(0) lgkmcnt(0)); one slot per launch, hipGraph of 128 launches replayed "
                    f"{args.replays} times", "device": torch.cuda.get_device_name(0), "rows": rows}
    json.dump(doc, open(args.out + ".json", "w"), indent=1)
    with open(args.out + ".txt", "w") as f:
        f.write("Kernel duration on the device clock (first wave entry -> last wave exit, measured by the waves; tools/kernel_device_clock.py)\n")
        f.write(doc["clock"] + "\n\n")
        f.write(f"{'config':<12}{'envs':>9} {'device clock us: mean':>22}{'median':>8}{'min':>7}{'p90':>7}{'max':>8} | {'launch period us':>17}{'(stamped)':>10}"
                f"{'boundary':>9} | {'rocprof avg us':>14} | frac of 8 TB/s on 8d bytes: device clock / launch period\n")
        for r in rows:
            d = r["device_clock_us"]
            rp = f"{r['rocprof_committed_avg_us']:.2f}" if r["rocprof_committed_avg_us"] else "-"
            f.write(f"{r['config']:<12}{r['envs']:>9} {d['mean']:>22.2f}{d['median']:>8.2f}{d['min']:>7.2f}{d['p90']:>7.2f}{d['max']:>8.2f} | "
                    f"{r['launch_period_us']:>17.2f}{r['launch_period_with_stamps_us']:>10.2f}{r['launch_boundary_us']:>9.2f} | {rp:>14} | "
                    + ("cache resident" if r["frac_of_8TBs_device_clock"] is None else f"{r['frac_of_8TBs_device_clock']:.3f}") +
                    f" / {r['frac_of_8TBs_launch_period']:.3f}   {r['kernel']}\n")
    print(open(args.out + ".txt").read())


if __name__ == "__main__":
    main()
