#!/bin/bash
# One GPU-box call that refreshes every committed measurement: GPU tests, the benchmark line, the kernel-trace
# summary of the same benchmark command and the PMC passes. usage: tools/round_refresh.sh <tag>
set -e
TAG=$1
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tee $OUT/pytest_gpu.log
echo "== bench"
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
echo "== kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- python3 bench.py --no-cpu-baseline --no-exact --no-extend > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
python3 tools/rocprof_summary.py $(dirname $(ls $OUT/kt/*/*kernel_stats.csv $OUT/kt/*kernel_stats.csv 2>/dev/null | head -n 1)) $OUT/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-exact --no-extend"
echo "== pmc"
tools/run_pmc.sh $TAG cornell_box 1024 1024 64 50 26
