#!/bin/bash
# Profiles of the bench.py command on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/prof/bench_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (one pass per counter, kernel trace only)
#                                                   -> gpurun_out/prof/pmc_<COUNTER>.csv
#   3. bench.py itself                              -> gpurun_out/prof/bench_n1.json
# tools/pmc_traffic.py turns (2) into profiles/traffic.json; copy what should be judged into profiles/.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qg_prof && mkdir -p /tmp/qg_prof
rocprofv3 --kernel-trace --stats -d /tmp/qg_prof/stats -o bench --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-parity --no-large-batch > "$OUT/stats_run.log" 2>&1
cp /tmp/qg_prof/stats/*kernel_stats.csv "$OUT/bench_kernel_stats.csv" 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace -d /tmp/qg_prof/$C -o pmc --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-parity --no-large-batch --steps 512 --warmup 64 > "$OUT/pmc_$C.log" 2>&1
    F=$(ls /tmp/qg_prof/$C/*counter_collection.csv 2>/dev/null | head -1)
    # keep the step kernel's dispatches only (the file is large): header + rows of the dominant kernel
    if [ -n "$F" ]; then (head -1 "$F"; grep "qm_step1_kernel<16, true, false>" "$F") > "$OUT/pmc_$C.csv"; fi
done
# 4. the same kernel at 2^20 envs (kernel duration >> launch boundary): the kernel-trace average and bench.py's own
#    per-launch figure agree there, which is the check that the 65 536-env gap is the profiler's per-dispatch serialisation
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/qg_prof/big -o bench --output-format csv -- python3 "$ROOT/bench.py" --envs 1048576 --steps 512 --warmup 64 --no-cpu-baseline --no-parity > "$OUT/bench_2p20_under_rocprof.json" 2> "$OUT/bench_2p20.err"
cp /tmp/qg_prof/big/*kernel_stats.csv "$OUT/bench_2p20_kernel_stats.csv" 2>/dev/null
cd "$ROOT" && python3 bench.py --envs 1048576 --steps 512 --warmup 64 --no-cpu-baseline --no-parity > "$OUT/bench_2p20.json" 2>> "$OUT/bench_2p20.err"
cd "$ROOT" && python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
ls -la "$OUT"
