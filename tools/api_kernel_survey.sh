#!/bin/bash
# rocprofv3 --kernel-trace --stats over tools/api_kernel_survey.py for a few env kinds; prints the top kernels of each.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/api_kernel_survey.txt"
for CASE in "clifford 16 65536 0" "clifford 16 65536 1" "clifford 24 65536 0" "linear_function 8 65536 1" "linear_function 24 65536 1" "permutation 12 65536 1" "pauli 20 65536 0"; do
  set -- $CASE
  export KIND=$1 N=$2 B=$3 INVERTS=$4
  rm -rf /tmp/aks
  rocprofv3 --kernel-trace --stats -d /tmp/aks -o run --output-format csv -- python3 "$ROOT/tools/api_kernel_survey.py" > /tmp/aks.log 2>&1
  echo "== $KIND ${N}q x $B envs, add_inverts=$INVERTS" >> "$OUT/api_kernel_survey.txt"
  python3 "$ROOT/tools/kernel_stats_top.py" /tmp/aks/*kernel_stats.csv 16 >> "$OUT/api_kernel_survey.txt" 2>&1
done
cat "$OUT/api_kernel_survey.txt"
