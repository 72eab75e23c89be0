#!/bin/bash
# PauliGym collector: plain run, then rocprofv3 --kernel-trace --stats of the same command; plus the first-layer kernel's own timing.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
python3 "$ROOT/tools/bench_collector_pauli.py" 2>/dev/null | tail -1 > "$OUT/collector_pauli.txt"
python3 "$ROOT/tools/bench_embed_words.py" 2>/dev/null | grep envs > "$OUT/embed_words.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/colp -o colp --output-format csv -- python3 "$ROOT/tools/bench_collector_pauli.py" > "$OUT/colp_prof.log" 2>&1
cp "$(ls /tmp/colp/*kernel_stats.csv | head -1)" "$OUT/colp_kernel_stats.csv"
python3 - <<PY >> "$OUT/collector_pauli.txt"
import csv
rows = list(csv.DictReader(open("$OUT/colp_kernel_stats.csv")))
print("rocprofv3 --kernel-trace --stats of the same command:")
for r in rows[:10]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us avg {float(r['Percentage']):6.2f} %")
PY
cat "$OUT/collector_pauli.txt" "$OUT/embed_words.txt"
