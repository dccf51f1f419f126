#!/bin/bash
# A/B of the compact sphere records on the million-sphere render: tools/ab_cloud.sh [spp]
SPP=${1:-16}
run() { timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact --no-extend --no-workloads --steps 2 --warmup 1 --scene sphere_cloud --width 3840 --height 2160 --spp $SPP --max-depth 10 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r['roofline']['stage_ms']
print('  ', r['config']['workload'].split(',')[0], '|', round(r['value'],1), 'Mpaths/s | seg/path', round(r['config']['segments_per_path'],2), '| pass kernel ms', round(o['shade_or_pass_kernel'],1), '| nan', round(r['config']['nan_pixel_fraction'],4))"; }
echo "== general records (PT_CFG_FAT_RECORDS)"; run --cfg-flags 8 || exit 1
echo "== compact sphere records over 64-byte nodes (PT_CFG_WIDE_NODES)"; run --cfg-flags 32 || exit 1
echo "== compact sphere records over 32-byte nodes (default)"; run || exit 1
