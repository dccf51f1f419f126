#!/bin/bash
# Render throughput on the non-headline configurations (bench lines are the Cornell box only): C1 RTiOW,
# C3 final scene / smoke, C4 one million spheres at reduced spp.
run() { timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact --steps 1 --warmup 1 "$@" | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r['roofline']['other']
print(r['config']['workload'], '|', round(r['value'],1), 'Mpaths/s | seg/path', round(r['config']['segments_per_path'],2), '| extend_ms', o['extend_ms_total'], 'shade_ms', o['shade_ms_total'], '| nan', r['config']['nan_pixel_fraction'])"; }
run --scene random_scene --width 1200 --height 800 --spp 100 &&
run --scene final_book_two --spp 256 &&
run --scene cornell_smoke --spp 256 &&
run --scene sphere_cloud --width 3840 --height 2160 --spp 16 --max-depth 10
