#!/bin/bash
# Per-phase cycle counts of the pool kernels from a diagnostic build (-DPT_POOL_TIMING): tools/pool_timing.sh <timing lib>
cp path-tracer_amd/libptgpu.so /tmp/base.so; cp "$1" path-tracer_amd/libptgpu.so
for a in "cornell_box 1024 1024 16 50 24" "random_scene 1200 800 16 50 24" "final_book_two 1024 1024 16 50 24" "cornell_smoke 1024 1024 16 50 24" "sphere_cloud 1920 1080 4 10 23"; do
  timeout -k 5 200 python3 tools/prof_driver.py $a 2>&1 | grep -v amdgpu.ids
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
