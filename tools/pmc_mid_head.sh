#!/bin/bash
# PMC counters of mid_head_sample_kernel (tools/bench_mid_head.py), one pass per counter group.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/pmc_mid_head"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pm$i -o pmc --output-format csv -- python3 "$ROOT/tools/bench_mid_head.py" > "$OUT/run$i.log" 2>&1
  F=$(ls /tmp/pm$i/*counter_collection.csv | head -1)
  (head -1 "$F"; grep "mid_head_sample_kernel" "$F" | head -120) > "$OUT/pmc$i.csv"
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/pmc*.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) >= 0:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[:20]  # the first batch size of the tool: 65 536 envs x 170 actions
        print(f"{k:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
