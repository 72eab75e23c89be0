#!/bin/bash
# GRBM_GUI_ACTIVE (GPU clock cycles the launch was in flight) per kernel / its duration from the kernel trace = the shader clock the launch ran at.  Development.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/clk
B=${B:-65536} rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/clk -o pmc --output-format csv -- python3 "$ROOT/tools/bench_auto_reset.py" > /tmp/clk.log 2>&1
python3 - <<'PY'
import csv, glob, collections
cyc = collections.defaultdict(list); dur = {}
for row in csv.DictReader(open(glob.glob("/tmp/clk/*kernel_trace.csv")[0])):
    dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row["Kernel_Name"])
for row in csv.DictReader(open(glob.glob("/tmp/clk/*counter_collection.csv")[0])):
    d = dur.get(row["Dispatch_Id"])
    if d and d[0] > 0:
        cyc[d[1][:70]].append((float(row["Counter_Value"]), d[0]))
for k, v in cyc.items():
    c = sum(x for x, _ in v) / len(v); ns = sum(y for _, y in v) / len(v)
    print(f"{k:72s} n={len(v):5d} cycles={c:10.0f} ns={ns:9.0f}  -> {c / ns * 1e3:7.0f} MHz (if the counter is one instance)")
PY
