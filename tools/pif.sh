#!/bin/bash
# Cornell throughput against the paths-in-flight budget (pass size), optionally under an environment setting:
# tools/pif.sh [VAR=value ...]
for p in 18 20 22 24 26; do
  env "$@" timeout -k 10 200 python3 bench.py --spp 64 --steps 3 --no-cpu-baseline --no-exact --paths-in-flight $((1<<p)) 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('$* 2^$p', round(r['value'],1), 'Mpaths/s')" || exit 1
done
