#!/bin/bash
# realtime 1080p frames per second with environment variants: tools/ab_rt.sh VAR=value ...
for kv in "$@"; do
  env $kv timeout -k 10 200 python3 bench.py --realtime --width 1920 --height 1080 --steps 300 --warmup 30 > /tmp/rt.json || exit 1
  python3 - "$kv" <<'PY'
import json, sys
r = json.load(open('/tmp/rt.json'))
print(sys.argv[1], round(r['config']['frames_per_s'], 1), 'frames/s', round(r['value'], 1), 'Mpaths/s')
PY
done
