#!/bin/bash
# Per-phase cycle counts of the pool kernels from a diagnostic build (-DPT_POOL_TIMING): tools/pool_timing.sh <timing lib> [scene ...]
cp path-tracer_amd/libptgpu.so /tmp/base.so; cp "$1" path-tracer_amd/libptgpu.so; shift
SCENES=${*:-cornell_box random_scene final_book_two cornell_smoke sphere_cloud}
for s in $SCENES; do
  case $s in
    random_scene) a="random_scene 1200 800 16 50 24";;
    sphere_cloud) a="sphere_cloud 1920 1080 4 10 23";;
    *) a="$s 1024 1024 16 50 24";;
  esac
  timeout -k 5 200 python3 tools/prof_driver.py $a 2>&1 | grep -v amdgpu.ids
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
