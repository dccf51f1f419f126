#!/bin/bash
# A/B of alternative builds of libptgpu.so on one bench workload: tools/ab_libs.sh "<bench args>" <lib> [<lib> ...]
# ("base" = the library in the tree). Prints throughput, segments per path, NaN fraction and the image's hash.
ARGS=$1; shift
run() { timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact --no-extend --no-workloads --steps 2 --warmup 1 $ARGS 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r['roofline']['stage_ms']
print('  ', r['config']['workload'].split(',')[0], '|', round(r['value'],1), 'Mpaths/s | seg/path', round(r['config']['segments_per_path'],3), '| pass kernel ms', round(o['shade_or_pass_kernel'],1), '| nan', round(r['config']['nan_pixel_fraction'],4), '| image', r['config']['image_sha1_16'])"; }
cp path-tracer_amd/libptgpu.so /tmp/base.so
for lib in "$@"; do
  if [ "$lib" = base ]; then cp /tmp/base.so path-tracer_amd/libptgpu.so; else cp "$lib" path-tracer_amd/libptgpu.so; fi
  echo "== $lib"
  run || { cp /tmp/base.so path-tracer_amd/libptgpu.so; exit 1; }
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
