#!/bin/bash
# rocprofv3 kernel statistics of the non-headline configurations (one short render each): tools/configs_trace.sh <out dir>
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o p -- python3 tools/prof_driver.py "$@" > $OUT/$name.log 2>&1 && python3 tools/rocprof_summary.py $OUT/$name $OUT/${name}_kernel_stats.csv "prof_driver.py $*" | head -8; }
run rtiow random_scene 1200 800 100 50 27 &&
run final final_book_two 1024 1024 64 50 26 &&
run cloud sphere_cloud 1920 1080 8 10 24 &&
run smoke cornell_smoke 1024 1024 64 50 26
