timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "fuzz or pass_form or realtime" 2>&1 | tail -2 || exit 1
for sc in cornell_box cornell_smoke; do
timeout -k 10 300 python3 bench.py --scene $sc --spp 128 --steps 3 --no-cpu-baseline --no-extend --no-exact 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('$sc', 'fast', round(r['value'],1))"
done
