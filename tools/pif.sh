#!/bin/bash
# Paths-in-flight sweep of the headline workload at 256 spp
for p in 26 27 28; do
  timeout -k 10 300 python3 bench.py --spp 1024 --steps 2 --no-cpu-baseline --no-exact --paths-in-flight $((1<<p)) > /tmp/pif.json || exit 1
  python3 - $p <<'PY'
import json, sys
r = json.load(open('/tmp/pif.json'))
print('2^'+sys.argv[1], round(r['value'], 1), 'Mpaths/s', round(r['ms_per_step'],1), 'ms/step')
PY
done
