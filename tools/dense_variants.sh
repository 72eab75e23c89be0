#!/bin/bash
# Development: live timings of the dense-observation variants (tools/build_variant.sh) + rocprofv3 kernel stats of the regular build.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r04"
mkdir -p "$OUT"
cd "$ROOT"
echo "base: $(python3 tools/bench_dense_obs.py)" > "$OUT/dense_variants.txt"
for L in qiskit_gym_amd/lib/variants/libqgym_*.so; do
    echo "$(basename $L): $(python3 tools/bench_dense_obs.py --lib $L --modes dense,tracked 2>/dev/null)" >> "$OUT/dense_variants.txt"
done
cat "$OUT/dense_variants.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qg_dense && mkdir -p /tmp/qg_dense
rocprofv3 --kernel-trace --stats -d /tmp/qg_dense -o run --output-format csv -- python3 "$ROOT/tools/bench_dense_obs.py" > /dev/null 2> "$OUT/dense_rocprof.err"
cp /tmp/qg_dense/*kernel_stats.csv "$OUT/dense_kernel_stats.csv" 2>/dev/null
head -12 "$OUT/dense_kernel_stats.csv"
