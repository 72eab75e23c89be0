#!/bin/bash
# FETCH_SIZE of tools/microbench_lines.hip's gather kernels (one rocprofv3 pass, counter + kernel trace only), next to the model it prints.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
hipcc -O3 --offload-arch=gfx950 "$ROOT/tools/microbench_lines.hip" -o /tmp/mb_lines || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qg_lines
/tmp/mb_lines > "$OUT/lines_model.txt"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/qg_lines -o pmc --output-format csv -- /tmp/mb_lines > /dev/null 2> "$OUT/lines_pmc.log"
F=$(ls /tmp/qg_lines/*counter_collection.csv | head -1)
python3 - "$F" "$OUT/lines_model.txt" > "$OUT/lines_microbench.txt" <<'PY'
import csv, sys, collections
vals = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if row["Counter_Name"] == "FETCH_SIZE" and "gather" in row["Kernel_Name"]:
        vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
print(open(sys.argv[2]).read())
print("PMC (rocprofv3 --pmc FETCH_SIZE, 2 x FETCH_SIZE x 1024 bytes per launch as MI355X_MICROARCH.md prescribes for gfx950; 2^20 envs, mean of the launches after the first):")
for k, v in vals.items():
    v = v[1:] if len(v) > 1 else v
    print(f"  {k:40s} {2 * sum(v) / len(v) * 1024 / (1 << 20):7.1f} B/env fetched")
PY
cat "$OUT/lines_microbench.txt"
