"""rocprofv3 PMC passes of bench.py (tools/profile_bench.sh) -> profiles/traffic.json.

HBM bytes per launch of the step kernel = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB):
MI355X_MICROARCH.md's HBM section: on gfx950 FETCH_SIZE counts half the bytes of a 16 B/lane read,
WRITE_SIZE is taken as is."""
import csv, json, os, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof")
KERNEL = "qm_step1_kernel<16, true, false>"
B, ALGO = 65536, 160


def read(counter):
    vals = []
    with open(os.path.join(src, f"pmc_{counter}.csv")) as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter and int(row["Grid_Size"]) == B:
                vals.append(float(row["Counter_Value"]))
    return vals


fetch, write = read("FETCH_SIZE"), read("WRITE_SIZE")
out = {
    "kernel": f"qg::{KERNEL} (CliffordGym 16q x {B} envs, one env.step())",
    "command": "tools/profile_bench.sh: rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-parity --steps 512 --warmup 64 (one pass per counter)",
    "FETCH_SIZE_KB_per_launch": statistics.mean(fetch),
    "WRITE_SIZE_KB_per_launch": statistics.mean(write),
    "correction": "MI355X_MICROARCH.md (HBM): FETCH_SIZE doubled on gfx950, WRITE_SIZE as is; both in KiB",
    "clifford_step_bytes_per_launch": (2 * statistics.mean(fetch) + statistics.mean(write)) * 1024,
    "algorithmic_bytes_per_launch": ALGO * B,
    "raw": {n: {"dispatches": len(v), "mean": statistics.mean(v), "min": min(v), "max": max(v)} for n, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write))},
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
