#!/bin/bash
# Round-3 extras (run through gpurun from the repo root): per-kernel durations of (1) the auto-reset legs of bench.py (the step kernel with its
# done list, compact_done, the reset kernel with scramble_tree) and (2) the multi-GPU code path on one rank (shard pack, ncclAllGather's copy,
# push_shard / wait_arrivals / release_window).  rocprofv3 --kernel-trace --stats only; the program itself after `--`.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof/r03"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qg_extra && mkdir -p /tmp/qg_extra
rocprofv3 --kernel-trace --stats -d /tmp/qg_extra/ar -o run --output-format csv -- python3 "$ROOT/bench.py" --profiling-run --no-cpu-baseline --no-parity --no-large-batch --no-configs --no-collector > /tmp/qg_extra/ar.json 2> /tmp/qg_extra/ar.log
python3 "$ROOT/tools/kernel_stats_top.py" /tmp/qg_extra/ar/*kernel_stats.csv 12 > "$OUT/auto_reset_kernel_stats.txt" 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/qg_extra/mg -o run --output-format csv -- python3 "$ROOT/bench.py" --force-multi --shard 3/8 --steps 20 --warmup 5 --profiling-run --no-cpu-baseline --no-parity --no-large-batch --no-default-config --no-configs --no-collector > /tmp/qg_extra/mg.json 2> /tmp/qg_extra/mg.log
python3 "$ROOT/tools/kernel_stats_top.py" /tmp/qg_extra/mg/*kernel_stats.csv 12 > "$OUT/multi_gpu_path_kernel_stats.txt" 2>&1
head -20 "$OUT/auto_reset_kernel_stats.txt" "$OUT/multi_gpu_path_kernel_stats.txt"
