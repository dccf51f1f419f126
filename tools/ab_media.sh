#!/bin/bash
# A/B of alternative builds of libptgpu.so on the scenes with media, both semantics: tools/ab_media.sh <lib> [<lib> ...] ("base" = the library in the tree)
run() { timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact --no-extend --no-workloads --steps 2 --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('  ', r['config']['workload'].split(',')[0].split(' ')[0], 'book' if r['config']['book_semantics'] else '', '|', round(r['value'],1), 'Mpaths/s | image', r['config']['image_sha1_16'])"; }
cp path-tracer_amd/libptgpu.so /tmp/base.so
for lib in "$@"; do
  if [ "$lib" = base ]; then cp /tmp/base.so path-tracer_amd/libptgpu.so; else cp "$lib" path-tracer_amd/libptgpu.so; fi
  echo "== $lib"
  run --scene final_book_two --spp 64 || { cp /tmp/base.so path-tracer_amd/libptgpu.so; exit 1; }
  run --scene final_book_two --scene-param 1 --spp 64 --book-semantics || exit 1
  run --scene cornell_smoke --spp 64 || exit 1
  run --scene cornell_smoke --scene-param 1 --spp 64 --book-semantics || exit 1
  run --scene cornell_smoke --scene-param 1 --spp 64 --book-semantics --cfg-flags 4 || exit 1
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
