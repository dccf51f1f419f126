#!/bin/bash
# Builds a variant of the library with extra compiler flags, beside the real one: tools/build_variant.sh <name> <flags...>
# -> build_ab/<name>.so (only ptgpu.hip is recompiled; the list pool object of the current build is reused)
set -e
cd "$(dirname "$0")/../path-tracer_amd"
NAME=$1; shift
mkdir -p ../build_ab/$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -mllvm -two-entry-phi-node-folding-threshold=8"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o ../build_ab/$NAME/ptgpu.o csrc/ptgpu.hip 2>/dev/null
/opt/rocm/bin/hipcc $FLAGS -shared -o ../build_ab/$NAME.so ../build_ab/$NAME/ptgpu.o csrc/pool_list.o csrc/flatten.cpp csrc/accel.cpp host/ptscene.cpp host/scenes.cpp host/capi.cpp -ldl
echo built build_ab/$NAME.so
