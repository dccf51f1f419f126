#!/bin/bash
# Render throughput on the non-headline configurations (bench lines are the Cornell box only): C1 RTiOW,
# C3 final scene / smoke in reference and book semantics, C4 one million spheres at reduced spp; the first three with
# the CPU oracle timed beside.
run() { timeout -k 10 400 python3 bench.py --no-exact --no-extend --steps 1 --warmup 1 "$@" | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r['roofline']['stage_ms']; c=r.get('cpu_baseline')
print(r['config']['workload'], '| book' if r['config']['book_semantics'] else '|', round(r['value'],1), 'Mpaths/s | seg/path', round(r['config']['segments_per_path'],2), '| pass kernel ms', round(o['shade_or_pass_kernel'],1), '| nan', round(r['config']['nan_pixel_fraction'],4),
      '| cpu', (str(round(c['value'],2))+' Mpaths/s on '+str(c['cores'])+' threads ('+c['sample']+'), gpu/cpu '+str(round(r['config']['gpu_over_cpu'],1))) if c else None)"; }
run --scene random_scene --width 1200 --height 800 --spp 100 &&
run --scene final_book_two --spp 256 &&
run --no-cpu-baseline --scene final_book_two --spp 256 --book-semantics &&
run --scene cornell_smoke --spp 256 &&
run --no-cpu-baseline --scene cornell_smoke --scene-param 1 --spp 256 --book-semantics &&
run --no-cpu-baseline --scene sphere_cloud --width 3840 --height 2160 --spp 16 --max-depth 10
