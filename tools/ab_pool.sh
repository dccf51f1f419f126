#!/bin/bash
# A/B of the two SAH traversal forms (a ray per thread / refilled lane pool) on renders: tools/ab_pool.sh "<env settings>" ...
run() { timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact --no-extend --steps 1 --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r['roofline']['stage_ms']
print('  ', r['config']['workload'].split(',')[0], '|', round(r['value'],1), 'Mpaths/s | extend_ms', round(o['extend'],1), 'shade_ms', round(o['shade_or_pass_kernel'],1))"; }
for cfg in "$@"; do
  echo "== $cfg"
  export $cfg
  run --scene sphere_cloud --width 3840 --height 2160 --spp 16 --max-depth 10 || exit 1
  run --scene final_book_two --spp 256 || exit 1
  run --scene random_scene --width 1200 --height 800 --spp 100 || exit 1
done
