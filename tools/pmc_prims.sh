#!/bin/bash
# VALU / SALU instruction counts of the all-miss experiment build on synthetic scenes: tools/pmc_prims.sh <lib>
set -e
cp "$1" path-tracer_amd/libptgpu.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "rect 0" "rect 8" "rect 16" "sphere 8" "box 8" "xbox 8"; do
  set -- $spec
  OUT=gpurun_out/pp_$1_$2
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/p -o p -- python3 tools/prof_prims.py $1 $2 > $OUT/log 2>&1
  python3 tools/pmc_summary.py $OUT/summary.csv $OUT/p > /dev/null
  echo "$spec: $(tail -1 $OUT/log | cut -c1-150)"
  grep k_shade $OUT/summary.csv
done
