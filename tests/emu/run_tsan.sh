#!/bin/bash
# Builds the emulated kernels with a sanitizer and runs tests/emu/tsan_driver.cpp.
# Usage: [SAN=thread|address,undefined] tests/emu/run_tsan.sh [f|sums|g|fb|fbc|rl|dc|helpers|voxel]
# (SAN defaults to thread; no argument: every kernel family)
set -e
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../thz_image_explorer_amd/csrc"
CXX="${CXX:-/opt/rocm/lib/llvm/bin/clang++}"
SAN="${SAN:-thread}"
"$CXX" -std=c++17 -O1 -g -fsanitize="$SAN" -fno-sanitize=float-divide-by-zero -DTHZ_EMU -I"$here" -I"$src" \
    -x c++ "$src/kernels.hip" "$src/voxel.hip" "$here/emu_harness.cpp" "$here/tsan_driver.cpp" -lpthread -lm -o "$here/tsan_driver"
ASAN_OPTIONS="detect_leaks=0 ${ASAN_OPTIONS}" TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=0 history_size=2 ${TSAN_OPTIONS}" "$here/tsan_driver" "$@"
