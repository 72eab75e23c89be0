#!/bin/bash
# PMC counters of embed_words_kernel (tools/bench_embed_words.py, PauliGym 20q x 65 536 envs), one pass per counter group.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/pmc_embed_words"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ONLY=${ONLY:-0} NEW_ONLY=1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pw$i -o pmc --output-format csv -- python3 "$ROOT/tools/bench_embed_words.py" > "$OUT/run$i.log" 2>&1
  F=$(ls /tmp/pw$i/*counter_collection.csv | head -1)
  (head -1 "$F"; grep "embed_words_kernel" "$F" | head -400) > "$OUT/pmc$i.csv"
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/pmc*.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
