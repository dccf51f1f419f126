#!/bin/bash
# The library's host sources (scene construction, flattening, SAH build, C ABI glue) under ASan + UBSan on the CPU:
# tools/host_sanitize.sh   (prints one line per recipe; any sanitizer report fails the run)
set -e
cd "$(dirname "$0")/../path-tracer_amd"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I../include -Icsrc -Ihost \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ csrc/flatten.cpp csrc/accel.cpp host/ptscene.cpp host/scenes.cpp host/capi.cpp \
    ../tools/host_sanitize.cpp -o /tmp/ptgpu_host_sanitize
/tmp/ptgpu_host_sanitize
