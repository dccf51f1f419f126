mkdir -p gpurun_out/s6
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 400 python3 bench.py > gpurun_out/s6/bench.json 2> gpurun_out/s6/bench.err || exit 1
tail -c 2500 gpurun_out/s6/bench.json
