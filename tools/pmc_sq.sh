#!/bin/bash
# usage: pmc_sq.sh <name> <run_config args...>   -- SQ counters of a config's step kernel (two passes of four counters).  Development.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -f /tmp/sq_$NAME.all.csv
for SET in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/sq_$NAME
  rocprofv3 --pmc $SET --kernel-trace -d /tmp/sq_$NAME -o pmc --output-format csv -- python3 "$ROOT/tools/run_config.py" "$@" > /tmp/sq_$NAME.log 2>&1
  F=$(ls /tmp/sq_$NAME/*counter_collection.csv | head -1)
  if [ -f /tmp/sq_$NAME.all.csv ]; then tail -n +2 "$F" >> /tmp/sq_$NAME.all.csv; else cp "$F" /tmp/sq_$NAME.all.csv; fi
done
python3 - /tmp/sq_$NAME.all.csv "$NAME" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "step" in k or "inv" in k:
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(sys.argv[2], k[:70])
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = m.get("SQ_WAVES", 1)
    print("   waves %.0f  wave_cycles/wave %.0f (x4 = shader cycles)  valu/wave %.0f salu/wave %.0f  wait_any %.2f  wait_inst %.2f  active_any %.2f active_valu %.2f (fractions of wave cycles)" % (
        w, m["SQ_WAVE_CYCLES"] / w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]))
PY
