#!/bin/bash
# One GPU-box call that refreshes the measurements of the non-headline configurations: render throughput with the CPU
# oracle beside (tools/configs.sh), counter passes of the tree scenes, the per-phase cycle table of the diagnostic
# build. usage: tools/round_refresh_configs.sh <tag> [<timing lib>]
TAG=$1
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
tools/configs.sh > $OUT/configs_render.txt 2> $OUT/configs_render.err || exit 1
cat $OUT/configs_render.txt
tools/run_pmc.sh ${TAG}_rtiow random_scene 1200 800 50 50 26 > $OUT/pmc_rtiow.txt 2>&1 || exit 1
tools/run_pmc.sh ${TAG}_final final_book_two 1024 1024 64 50 26 > $OUT/pmc_final.txt 2>&1 || exit 1
tools/run_pmc.sh ${TAG}_smoke cornell_smoke 1024 1024 64 50 26 > $OUT/pmc_smoke.txt 2>&1 || exit 1
tools/run_pmc.sh ${TAG}_cloud sphere_cloud 1920 1080 16 10 25 > $OUT/pmc_cloud.txt 2>&1 || exit 1
tools/pmc_mem.sh ${TAG}_cloud sphere_cloud 1920 1080 16 10 25 > $OUT/pmcmem_cloud.txt 2>&1 || exit 1
if [ -n "$2" ]; then tools/pool_timing.sh $2 > $OUT/pool_timing.txt 2>&1 || exit 1; fi
