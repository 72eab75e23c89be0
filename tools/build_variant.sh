#!/bin/bash
# Development aid: libqgym with ONE translation unit rebuilt under extra -D flags, as qiskit_gym_amd/lib/variants/libqgym_<name>.so
# (the other objects come from the regular build; run `make -C qiskit_gym_amd/csrc` first).  tools/*.py --lib <path> load it.
#   tools/build_variant.sh <name> <source.hip> <flags...>
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; SRC="$2"; shift 2
cd "$ROOT/qiskit_gym_amd/csrc"
mkdir -p build/variants ../lib/variants
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
$HIPCC -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function --offload-arch=gfx950 "$@" -x hip -c "$SRC" -o "build/variants/$SRC.$NAME.o"
OBJS=$(for f in $(sed -n 's/^SRCS = //p' Makefile); do [ "$f" = "$SRC" ] || echo "build/$f.o"; done)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "../lib/variants/libqgym_$NAME.so" $OBJS "build/variants/$SRC.$NAME.o" -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined -ldl
echo "built qiskit_gym_amd/lib/variants/libqgym_$NAME.so"
