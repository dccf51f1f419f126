#!/bin/bash
# Realtime 1080p frame rates of every scene and of the presentation ring, as one JSON object on stdout:
# tools/realtime_rates.sh > gpurun_out/realtime.json   (on the GPU box, from the repository root)
set -e
run() { key=$1; shift; timeout -k 10 300 python3 bench.py --realtime --width 1920 --height 1080 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); c=r['config']
print(json.dumps({'$key': {'frames_per_s': c['frames_per_s'], 'Mpaths_s': r['value'], 'workload': c['workload'], 'present': c['present']}}))"; }
{
for s in cornell_box cornell_smoke final_book_two random_scene; do run $s --scene $s --steps 1000 --warmup 100; done
run present_ring --steps 1000 --warmup 50 --present ring
run present_ring8 --steps 1000 --warmup 50 --present ring8
} | python3 -c "
import json,sys
out={'command': 'python3 bench.py --realtime --scene <scene> --width 1920 --height 1080 --steps 1000 --warmup 100 [--present ring|ring8 --warmup 50]; frames are issued with PT_RENDER_NO_WAIT and the timed region ends with a device synchronisation', 'results': {}}
for l in sys.stdin: out['results'].update(json.loads(l))
print(json.dumps(out, indent=1))"
