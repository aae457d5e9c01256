#!/bin/bash
# Host-side AddressSanitizer pass over the C ABI layer.
#   make -C mppi_numba_amd/csrc asan          -> build/libmppi_asan.so (host code instrumented)
#   bash tools/asan_abi.sh [pytest args]      -> the ABI / handle-lifetime / error-path tests under it
# Covers library load, symbol export, struct layouts and every error path that is taken before a
# device is touched.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/build/libmppi_asan.so
[ -f "$LIB" ] || make -C "$ROOT/mppi_numba_amd/csrc" asan || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export MPPI_HIP_LIB=$LIB
export LD_PRELOAD=$RT
# (the interpreter and the HIP runtime are not instrumented: their one-time allocations are not leaks of ours)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_odr_violation=0
cd "$ROOT"
# (The HIP runtime of this image cannot be initialised under the sanitizer -- hipGetDeviceCount dies inside
#  libamdhip64 with an out-of-memory report, and there is no sanitizer build of the runtime under
#  /opt/rocm/lib/asan -- so on a GPU box, too, the pass is over the tests that touch no device:
#  profiles/r03_asan.md.)
TESTS="tests/test_host_and_abi.py"
MARK="-m not\ gpu"
if [ $# -gt 0 ]; then TESTS="$@"; fi
eval python -m pytest $TESTS -q -x -p no:cacheprovider $MARK
