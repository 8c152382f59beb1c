#!/bin/bash
# Builds the rejected kernel variants (lab notes) into tools/experiments/libnvmk_experiments.so.
# Not part of the product build: __graft_entry__.build() never touches this directory.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$ROOT/include" -Wno-unused-function \
  -x hip "$HERE/fp4_experiments.hip" "$ROOT/nvmolkit_amd/csrc/runtime.cpp" \
  -o "$HERE/libnvmk_experiments.so" -Wl,-rpath,/opt/rocm/lib
echo "$HERE/libnvmk_experiments.so"
