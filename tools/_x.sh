mkdir -p gpurun_out/rt
for sc in cornell_box:cornell random_scene:rtiow final_book_two:final cornell_smoke:smoke; do
  s=${sc%%:*}; t=${sc##*:}
  timeout -k 10 200 python3 bench.py --scene $s --realtime --width 1920 --height 1080 --steps 300 --warmup 30 > gpurun_out/rt/$t.json 2>/dev/null || exit 1
  python3 -c "
import json;r=json.load(open('gpurun_out/rt/$t.json'));print('$t', round(r['config']['frames_per_s'],1),'fps',round(r['value'],1))"
done
