#!/bin/bash
# SQ / SQC counters of the one-launch auto-reset kernel (tools/bench_auto_reset.py): instructions, wait fractions, instruction cache.  Development.
# SETS="A B;C D" overrides the counter sets (one pass per set).
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra SETS_A <<< "${SETS:-SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES;SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU;SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE;SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_TC_STALL;SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE}"
for SET in "${SETS_A[@]}"; do
  rm -rf /tmp/sq_reset
  B=${B:-65536} rocprofv3 --pmc $SET --kernel-trace -d /tmp/sq_reset -o pmc --output-format csv -- python3 "$ROOT/tools/bench_auto_reset.py" > /tmp/sq_reset.log 2>&1
  F=$(ls /tmp/sq_reset/*counter_collection.csv | head -1)
  python3 - "$F" "${KERNEL:-reset_step}" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if sys.argv[2] in k:
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k[:60], {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
done
