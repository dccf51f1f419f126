#!/bin/bash
# A/B of alternative builds of libptgpu.so on the tree-scene renders: tools/ab_configs.sh <lib> [<lib> ...]
run() { timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact --no-extend --steps 2 --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r['roofline']['stage_ms']
print('  ', r['config']['workload'].split(',')[0], 'book' if r['config']['book_semantics'] else '', '|', round(r['value'],1), 'Mpaths/s | seg/path', round(r['config']['segments_per_path'],2), '| nan', round(r['config']['nan_pixel_fraction'],3))"; }
cp path-tracer_amd/libptgpu.so /tmp/base.so
for lib in base "$@"; do
  if [ "$lib" = base ]; then cp /tmp/base.so path-tracer_amd/libptgpu.so; else cp "$lib" path-tracer_amd/libptgpu.so; fi
  echo "== $lib"
  run --scene random_scene --width 1200 --height 800 --spp 100 || exit 1
  run --scene final_book_two --spp 64 || exit 1
  run --scene final_book_two --scene-param 1 --spp 64 --book-semantics || exit 1
  run --scene cornell_smoke --spp 64 || exit 1
  run --scene sphere_cloud --width 3840 --height 2160 --spp 8 --max-depth 10 || exit 1
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
