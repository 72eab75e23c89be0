#!/bin/bash
# The CPU test suite against the sanitizer build of libqgym's HOST side (make -C qiskit_gym_amd/csrc asan: -fsanitize=address,undefined on every
# translation unit's host half) and of the oracle (make -C oracle libqgym_oracle_asan.so).  CPU container only: never on the GPU box (GPU
# AddressSanitizer is not available on this pool; the box never loads these libraries).  Output: profiles/<round>/host_asan.txt.
#   ROUND=r05 bash tools/run_cpu_tests_asan.sh
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
ROUND="${ROUND:-r05}"
OUT="$ROOT/profiles/$ROUND/host_asan.txt"
mkdir -p "$(dirname "$OUT")"
make -C "$ROOT/qiskit_gym_amd/csrc" asan -j6 > /tmp/qg_asan_build.log 2>&1 || { tail -20 /tmp/qg_asan_build.log; exit 1; }
make -C "$ROOT/oracle" libqgym_oracle_asan.so > /dev/null 2>&1 || exit 1
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
cd "$ROOT"
{
    echo "CPU tests against lib/libqgym_asan.so (host code: -fsanitize=address,undefined; $(date -u +%Y-%m-%d))"
    echo "runtime: $RT; tests: test_abi.py test_dispatch.py test_distributed_cpu.py (+ the oracle's golden / physics / symmetry tests on its own sanitizer build)"
    echo
    # detect_leaks=0: CPython itself is not leak-clean; halt_on_error=1 + abort: a finding fails the run.  The oracle's gcc build uses libasan from gcc:
    # its tests run in their own process with that runtime.
    LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 QGYM_LIB_ASAN=1 \
        python -m pytest tests/test_abi.py tests/test_dispatch.py tests/test_distributed_cpu.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
    echo "exit code of the libqgym run: ${PIPESTATUS[0]}"
    echo
    LD_PRELOAD="$(gcc -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 QGYM_ORACLE_ASAN=1 \
        python -m pytest tests/test_oracle_golden.py tests/test_oracle_symmetry.py tests/test_physics.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -5
    echo "exit code of the oracle run: ${PIPESTATUS[0]}"
} > "$OUT" 2>&1
cat "$OUT"
