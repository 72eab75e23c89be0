#!/bin/bash
# CliffordGym 16q collector at 65 536 envs under rocprofv3 --kernel-trace --stats (per-kernel split of a collection step).
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ONLY="${ONLY:-65536}"
rocprofv3 --kernel-trace --stats -d /tmp/col -o col --output-format csv -- python3 "$ROOT/tools/bench_collector.py" > "$OUT/col_prof.log" 2>&1
cp "$(ls /tmp/col/*kernel_stats.csv | head -1)" "$OUT/col_kernel_stats.csv"
python3 - <<PY > "$OUT/collector_kernel_stats.txt"
import csv
rows = list(csv.DictReader(open("$OUT/col_kernel_stats.csv")))
print("rocprofv3 --kernel-trace --stats -- python3 tools/bench_collector.py (ONLY=$ONLY: CliffordGym 16q x $ONLY envs, bf16 BasicPolicy 1024-512-256-{170,1}, packed rollout observation)")
for r in rows[:10]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us avg {float(r['Percentage']):6.2f} %")
PY
cat "$OUT/collector_kernel_stats.txt"; tail -1 "$OUT/col_prof.log"
