#!/bin/bash
# A/B timing of alternative builds of libptgpu.so on one box: tools/ab.sh <lib> [<lib> ...]
# Each library is copied over path-tracer_amd/libptgpu.so in the box's scratch copy and the short bench is run.
set -e
cp path-tracer_amd/libptgpu.so /tmp/base.so
for lib in base "$@"; do
  if [ "$lib" = base ]; then cp /tmp/base.so path-tracer_amd/libptgpu.so; else cp "$lib" path-tracer_amd/libptgpu.so; fi
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --spp 64 --steps 3 --no-cpu-baseline --no-exact --no-extend > /tmp/ab.json
    python3 - "$lib" <<'PY'
import json, sys
r = json.load(open('/tmp/ab.json'))
o = r['roofline']['stage_ms']
print(sys.argv[1], round(r['value'], 1), 'Mpaths/s  shade_ms', round(o['shade_or_pass_kernel'], 2), 'seg/path', round(r['config']['segments_per_path'], 3), 'nan', r['config']['nan_pixel_fraction'])
PY
  done
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
