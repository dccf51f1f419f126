#!/bin/bash
# Like ab.sh, with an environment setting per run: tools/ab_env.sh <lib|base> VAR=value [VAR=value ...]
set -e
LIB=$1; shift
cp path-tracer_amd/libptgpu.so /tmp/base.so
[ "$LIB" != base ] && cp "$LIB" path-tracer_amd/libptgpu.so
for kv in "$@"; do
  env $kv timeout -k 10 200 python3 bench.py --spp 64 --steps 3 --no-cpu-baseline --no-exact --no-extend > /tmp/ab.json
  python3 - "$LIB $kv" <<'PY'
import json, sys
r = json.load(open('/tmp/ab.json'))
o = r['roofline']['stage_ms']
print(sys.argv[1], round(r['value'], 1), 'Mpaths/s  shade_ms', round(o['shade_or_pass_kernel'], 2))
PY
done
cp /tmp/base.so path-tracer_amd/libptgpu.so
