"""rocprofv3 PMC passes of tools/profile_bench.sh -> profiles/r05/traffic.json (and the kernel-stats files copied beside it).
  python tools/pmc_traffic.py [src = gpurun_out/prof] [dst = profiles/r05]

HBM bytes per launch of a step kernel = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): MI355X_MICROARCH.md's HBM
section -- on gfx950 FETCH_SIZE counts half the bytes of a 16 B/lane read, WRITE_SIZE is taken as is; each counter is
collected in its own pass (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass)."""
import csv
import glob
import itertools
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r05")
os.makedirs(dst, exist_ok=True)

# run name -> (kernel name PREFIX -- trailing template arguments vary by call site --, envs, SURVEY 8(d) bytes per env-step, bytes the
# kernel must move per env-step)
RUNS = {
    "bench": ("qm_step1_kernel<16, true, false", 65536, 160, 90),
    "C3_1048576": ("qm_step1_kernel<16, true, false", 1048576, 160, 90),
    "C3_4194304": ("qm_step1_kernel<16, true, false", 4194304, 160, 90),
    # two lanes per env: Grid_Size is 2 x envs.  Needed: 128 B of state read; the envs whose coin fired (half) rewrite all 128 B, the others the
    # gate's <= 2 groups (~24 B): 76 B on average; action 4, coin 1, depth 4 + 4, flags 1 + 1, log lengths 8 + 8, log entry 4, reward 4, done 1, success 1
    "C3d": ("qm_inv2_kernel<16, true", 2 * 65536, 160, 245),
    "C3d_4194304": ("qm_inv2_kernel<16, true", 2 * 4194304, 160, 245),
    "C2": ("word_step_kernel<false>", 8192, 32, 32),
    # two 12-byte qubit records read and written, the 64-byte rotation block's touched 16-bit slices + bookkeeping, 22 B of scalars
    "C5": ("ptile_step1c_kernel<20, 8", 65536, 494, 130),
    "C5_1048576": ("ptile_step1c_kernel<20, 8", 1048576, 494, 130),
    "C5_4194304": ("ptile_step1c_kernel<20, 8", 4194304, 494, 130),
    # SURVEY 8d's dense-observation mode (tools/bench_dense_obs.py): the full rewrite (1 KiB written, 128 B read per env), the step that keeps a
    # resident dense observation current (8d counts 1 184 B; it has to move the step's 90 B + ~2 changed rows of 32 B), and the same with the
    # reference-default options (two lanes per env: half of the envs rewrite 1 KiB, the others the gate's rows)
    "dense": ("qm_dense_stream_kernel<2>", 65536, 1152, 1152),
    "tracked": ("qm_step1_kernel<16, true, false, false, true>", 65536, 1184, 154),
    "tracked_default": ("qm_inv2_kernel<16, false, false, true>", 2 * 65536, 1184, 245 + 512 + 48 - 20),  # (tools/bench_dense_obs.py --inverts keeps no solution log)
}


def read(run, counter, kernel, envs):
    path = os.path.join(src, f"pmc_{run}_{counter}.csv")
    vals = []
    if not os.path.exists(path):
        return vals
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter and int(row["Grid_Size"]) >= envs:
                vals.append(float(row["Counter_Value"]))
    return vals


def stats(run, kernel):
    path = os.path.join(src, f"{run}_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    calls, total, mins, maxs = 0, 0.0, [], []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Name"]:  # every instantiation with this prefix, calls-weighted
                calls += int(row["Calls"])
                total += float(row["AverageNs"]) * int(row["Calls"])
                mins.append(float(row["MinNs"]))
                maxs.append(float(row["MaxNs"]))
    if not calls:
        return None
    return {"calls": calls, "avg_us": total / calls / 1e3, "min_us": min(mins) / 1e3, "max_us": max(maxs) / 1e3}


out = {"method": "tools/profile_bench.sh: rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 <bench.py | tools/run_config.py ...>, "
                 "one pass per counter; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE doubled on gfx950, "
                 "WRITE_SIZE as is, both in KiB); mean over the step kernel's dispatches",
       "by_envs": {}, "configs": {}}
try:  # a run of tools/profile_bench.sh in two parts: keep what the other part measured
    prev = json.load(open(os.path.join(dst, "traffic.json")))
    out["by_envs"].update(prev.get("by_envs", {}))
    out["configs"].update(prev.get("configs", {}))
except Exception:
    pass
for run, (kernel, envs, algo, needed) in RUNS.items():
    fetch, write = read(run, "FETCH_SIZE", kernel, envs), read(run, "WRITE_SIZE", kernel, envs)
    if not fetch or not write:
        continue
    fb, wb = 2 * statistics.mean(fetch) * 1024, statistics.mean(write) * 1024
    st = stats(run, kernel)
    live = None
    lp = os.path.join(src, f"{run}_live.json")
    if os.path.exists(lp):
        try:
            live = json.loads(open(lp).read().strip().splitlines()[-1])
        except Exception:
            live = None
    if run in ("C3d", "C3d_4194304", "tracked_default"):
        envs //= 2
    e = {"kernel": "qg::" + kernel + "...>", "envs": envs, "fetch_bytes": fb, "write_bytes": wb, "bytes_per_launch": fb + wb,
         "bytes_per_env": (fb + wb) / envs, "fetch_per_env": fb / envs, "write_per_env": wb / envs,
         "survey_8d_bytes_per_env": algo, "needed_bytes_per_env": needed,
         "dispatches": {"FETCH_SIZE": len(fetch), "WRITE_SIZE": len(write)},
         "raw_KiB": {"FETCH_SIZE": {"mean": statistics.mean(fetch), "min": min(fetch), "max": max(fetch)},
                     "WRITE_SIZE": {"mean": statistics.mean(write), "min": min(write), "max": max(write)}},
         "rocprof_kernel_stats": st, "live": live}
    if needed:
        e["traffic_over_needed"] = (fb + wb) / envs / needed
    if st:
        e["rocprof_GBs_algorithmic"] = algo * envs / st["avg_us"] / 1e3
        e["rocprof_frac_algorithmic"] = e["rocprof_GBs_algorithmic"] / 8000
        e["rocprof_GBs_moved"] = (fb + wb) / st["avg_us"] / 1e3
        e["rocprof_frac_moved"] = e["rocprof_GBs_moved"] / 8000
        if needed:
            e["rocprof_frac_needed"] = needed * envs / st["avg_us"] / 1e3 / 8000
    out["configs"][run] = e
    if kernel.startswith("qm_step1_kernel") and run != "tracked":
        out["by_envs"][str(envs)] = e
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
copied = []
for f in glob.glob(os.path.join(src, "*kernel_stats.csv")):
    name = os.path.basename(f)
    name = {"C3_1048576_kernel_stats.csv": "bench_1048576_kernel_stats.csv", "C3_4194304_kernel_stats.csv": "bench_4194304_kernel_stats.csv"}.get(name, name)
    shutil.copy(f, os.path.join(dst, name))
    copied.append(name)
for f in glob.glob(os.path.join(src, "pmc_*_SIZE.csv")):  # first 40 dispatches of each PMC pass as evidence (the full files are large)
    with open(f) as fh:
        head = list(itertools.islice(fh, 41))
    open(os.path.join(dst, os.path.basename(f).replace(".csv", "_first40.csv")), "w").writelines(head)
for f in ("bench_n1.json", "bench_driver_args.json", "bench_under_rocprof.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
print(json.dumps({k: {"B/env": round(v["bytes_per_env"], 1), "fetch": round(v["fetch_per_env"], 1), "write": round(v["write_per_env"], 1),
                      "rocprof_avg_us": v["rocprof_kernel_stats"]["avg_us"] if v["rocprof_kernel_stats"] else None,
                      "frac_moved": round(v.get("rocprof_frac_moved", 0), 3), "frac_8d": round(v.get("rocprof_frac_algorithmic", 0), 3)}
                  for k, v in out["configs"].items()}, indent=1))
print("copied:", sorted(copied))
