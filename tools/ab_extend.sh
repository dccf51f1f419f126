#!/bin/bash
# A/B of alternative libraries on the extend batches: tools/ab_extend.sh <scene> <lib> [<lib> ...]
set -e
SCENE=$1; shift
cp path-tracer_amd/libptgpu.so /tmp/base.so
for lib in base "$@"; do
  if [ "$lib" = base ]; then cp /tmp/base.so path-tracer_amd/libptgpu.so; else cp "$lib" path-tracer_amd/libptgpu.so; fi
  timeout -k 10 300 python3 tools/bench_extend.py $SCENE 2>/dev/null | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('$lib', r['scene'], r['batch'], round(r['gpu_Mrays_s'],1), 'Mrays/s ids', r['id_agreement'])"
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
