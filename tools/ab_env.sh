#!/bin/bash
# A/B of an environment switch on one bench workload: tools/ab_env.sh "<bench args>" VAR=value [VAR=value ...]
# (the first run is without any of them)
ARGS=$1; shift
run() { env "$@" timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact --no-extend --no-workloads --steps 2 --warmup 1 $ARGS 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r['roofline']['stage_ms']
print('  ', r['config']['workload'].split(',')[0], '|', round(r['value'],1), 'Mpaths/s | seg/path', round(r['config']['segments_per_path'],3), '| pass kernel ms', round(o['shade_or_pass_kernel'],1), '| nan', round(r['config']['nan_pixel_fraction'],4), '| image', r['config']['image_sha1_16'])"; }
echo "== default"; run X_UNUSED=1 || exit 1
for kv in "$@"; do echo "== $kv"; run "$kv" || exit 1; done
echo "== default"; run X_UNUSED=1
