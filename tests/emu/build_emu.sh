#!/bin/bash
# Builds the host-thread emulation of the kernels (test infrastructure only).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../thz_image_explorer_amd/csrc"
CXX="${CXX:-/opt/rocm/lib/llvm/bin/clang++}"
[ -x "$CXX" ] || CXX=g++
"$CXX" -std=c++17 -O1 -g -DTHZ_EMU -fPIC -shared -I"$here" -I"$src" \
    -x c++ "$src/kernels.hip" "$src/voxel.hip" "$here/emu_harness.cpp" -lpthread -lm -o "$here/libthz_emu.so"
