#!/bin/bash
# Profiles of a round (run through gpurun from the repo root: ROUND=r05 bash tools/profile_bench.sh; results land in gpurun_out/prof,
# tools/pmc_traffic.py turns them into what is committed under profiles/$ROUND/):
#   1. rocprofv3 --kernel-trace --stats of the bench.py command the driver runs       -> bench_kernel_stats.csv
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (one per counter, kernel trace only) of the same kernel at 65 536 envs
#   3. kernel stats + PMC passes of every other configuration's step kernel: C2 (word_step_kernel), C5 (ptile_step1c_kernel),
#      C3 at 2^20 and 2^22 envs (beyond the Infinity Cache), C3 with the reference-default options (also at 2^22), C5 at 2^20 and 2^22
#   4. the clocks no profiler is involved in: tools/kernel_device_clock.py, tools/auto_reset_breakdown.py, tools/bench_word_reset.py
# Each rocprofv3 command has the program itself after `--` (python3 <script>), never a shell or env wrapper.
set -u
ROUND="${ROUND:-r05}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qg_prof && mkdir -p /tmp/qg_prof
PART="${PART:-all}"   # A: the bench.py command and the configurations; B: dense observation, auto-reset, clocks, the bench lines (needs A's files); all: both
BENCH_ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-large-batch --no-default-config --no-configs --no-collector --no-dense-obs --profiling-run"
pmc_pass() {  # name, counter, script args...
    local name="$1" counter="$2"; shift 2
    rocprofv3 --pmc "$counter" --kernel-trace -d "/tmp/qg_prof/${name}_$counter" -o pmc --output-format csv -- python3 "$@" > "$OUT/pmc_${name}_$counter.log" 2>&1
    local F
    F=$(ls /tmp/qg_prof/${name}_$counter/*counter_collection.csv 2>/dev/null | head -1)
    # keep the step kernels' dispatches only (the files are large): header + rows of *_step* kernels
    if [ -n "$F" ]; then (head -1 "$F"; grep -E "step1?c?_kernel|inv2_kernel|dense_stream_kernel" "$F") > "$OUT/pmc_${name}_$counter.csv"; fi
}
stats_pass() {  # name, script args...
    local name="$1"; shift
    rocprofv3 --kernel-trace --stats -d "/tmp/qg_prof/${name}_stats" -o run --output-format csv -- python3 "$@" > "$OUT/${name}_under_rocprof.json" 2> "$OUT/${name}_stats.log"
    cp /tmp/qg_prof/${name}_stats/*kernel_stats.csv "$OUT/${name}_kernel_stats.csv" 2>/dev/null
}
if [ "$PART" != "B" ]; then
echo "== bench.py kernel stats" && date
rocprofv3 --kernel-trace --stats -d /tmp/qg_prof/stats -o bench --output-format csv -- python3 "$ROOT/bench.py" $BENCH_ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats_run.log"
cp /tmp/qg_prof/stats/*kernel_stats.csv "$OUT/bench_kernel_stats.csv" 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
    echo "== bench.py pmc $C" && date
    pmc_pass bench $C "$ROOT/bench.py" --gpus 1 --steps 512 --warmup 64 --no-cpu-baseline --no-parity --no-large-batch --no-default-config --no-configs --no-collector --no-dense-obs --profiling-run
done
for CFG in "C2" "C5" "C3d" "C3 --envs 1048576" "C3 --envs 4194304" "C5 --envs 1048576" "C5 --envs 4194304" "C3d --envs 4194304"; do
    set -- $CFG
    NAME=$(echo "$CFG" | tr -d ' -' | sed 's/envs/_/')
    echo "== run_config $CFG ($NAME)" && date
    python3 "$ROOT/tools/run_config.py" --config "$@" > "$OUT/${NAME}_live.json" 2> "$OUT/${NAME}_live.err"
    stats_pass "$NAME" "$ROOT/tools/run_config.py" --config "$@" --steps 512
    for C in FETCH_SIZE WRITE_SIZE; do
        pmc_pass "$NAME" $C "$ROOT/tools/run_config.py" --config "$@" --steps 256
    done
done
fi
mkdir -p "$OUT/$ROUND"
cp "$ROOT/profiles/$ROUND/"* "$OUT/$ROUND/" 2>/dev/null   # (PART=B on a fresh box: part A's summaries, committed in between; traffic.json is merged into)
if [ "$PART" = "A" ]; then
    python3 "$ROOT/tools/pmc_traffic.py" "$OUT" "$OUT/$ROUND" > "$OUT/pmc_traffic.log" 2>&1
    ls -la "$OUT" "$OUT/$ROUND"; exit 0
fi
echo "== SURVEY 8d's dense-observation mode: full rewrite, tracked, tracked with the reference-default options" && date
python3 "$ROOT/tools/bench_dense_obs.py" > "$OUT/dense_live.json" 2> "$OUT/dense_live.err"
python3 "$ROOT/tools/bench_dense_obs.py" --inverts > "$OUT/dense_default_live.json" 2>> "$OUT/dense_live.err"
for RUN in "dense --modes dense" "tracked --modes tracked" "tracked_default --modes tracked --inverts"; do
    set -- $RUN
    NAME=$1; shift
    stats_pass "$NAME" "$ROOT/tools/bench_dense_obs.py" "$@"
    for C in FETCH_SIZE WRITE_SIZE; do
        pmc_pass "$NAME" $C "$ROOT/tools/bench_dense_obs.py" "$@" --replays 2
    done
done
echo "== desynchronised auto-reset (one launch per pair, and two) and reset_done alone" && date
stats_pass auto_reset "$ROOT/tools/bench_auto_reset.py"
python3 "$ROOT/tools/bench_auto_reset.py" > "$OUT/auto_reset_live.txt" 2>&1
B=32768 python3 "$ROOT/tools/bench_auto_reset.py" >> "$OUT/auto_reset_live.txt" 2>&1
B=131072 python3 "$ROOT/tools/bench_auto_reset.py" >> "$OUT/auto_reset_live.txt" 2>&1
UNFUSED=1 python3 "$ROOT/tools/bench_auto_reset.py" >> "$OUT/auto_reset_live.txt" 2>&1
python3 "$ROOT/tools/bench_reset_done.py" > "$OUT/reset_done_live.txt" 2>&1
python3 "$ROOT/tools/bench_word_reset.py" > "$OUT/word_reset_live.txt" 2>&1
python3 "$ROOT/tools/auto_reset_breakdown.py" --out "$OUT/auto_reset_breakdown.txt" > /dev/null 2>&1
echo "== post-processing on the box: traffic.json + the files bench.py reads" && date
mkdir -p "$OUT/$ROUND"
python3 "$ROOT/tools/pmc_traffic.py" "$OUT" "$OUT/$ROUND" > "$OUT/pmc_traffic.log" 2>&1
cp "$OUT/dense_live.json" "$OUT/dense_default_live.json" "$OUT/auto_reset_live.txt" "$OUT/reset_done_live.txt" "$OUT/word_reset_live.txt" "$OUT/auto_reset_breakdown.txt" "$OUT/$ROUND/" 2>/dev/null
mkdir -p "$ROOT/profiles/$ROUND" && cp "$OUT/$ROUND/"* "$ROOT/profiles/$ROUND/"   # the box's copy of the repo: bench.py below reads them
echo "== the kernel device clock beside the rocprofv3 averages made above" && date
python3 "$ROOT/tools/kernel_device_clock.py" --out "$OUT/$ROUND/kernel_device_clock" --profile-dir "$OUT/$ROUND" > "$OUT/kernel_device_clock.log" 2>&1
cp "$OUT/$ROUND/kernel_device_clock."* "$ROOT/profiles/$ROUND/" 2>/dev/null
echo "== bench.py plain (reads the summaries made above)" && date
cd "$ROOT" && python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/$ROUND/bench_driver_args.json" 2> "$OUT/bench_driver_args.err"
cd "$ROOT" && python3 bench.py > "$OUT/$ROUND/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "== the multi-GPU code path on one rank (qg_comm: RCCL + direct write cadences)" && date
cd "$ROOT" && python3 bench.py --force-multi --shard 3/8 --steps 20 --warmup 5 --no-cpu-baseline --no-large-batch --no-default-config --no-configs --no-collector > "$OUT/$ROUND/bench_force_multi_rank3of8.json" 2> "$OUT/bench_force_multi.err"
cd "$ROOT" && python3 bench.py --gpus 2 --handover direct --ranks-share-gpu0 --steps 20 --warmup 5 --no-cpu-baseline --no-large-batch --no-default-config --no-configs --no-collector > "$OUT/$ROUND/bench_two_ranks_one_gpu_direct.json" 2> "$OUT/bench_two_ranks.err"
ls -la "$OUT" "$OUT/$ROUND"
