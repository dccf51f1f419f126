#!/bin/bash
# one counter pass: instruction-cache and instruction-fetch counters of the pool kernels
set -e
OUT=gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 60 rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -i -E "ICACHE|IFETCH|INST_LEVEL|WAIT_INST" $OUT/avail.txt | head -40 > $OUT/avail_grep.txt || true
run() { name=$1; args=$2; shift; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/prof_driver.py $args > $OUT/$name.log 2>&1; }
run final "final_book_two 1024 1024 64 50 26 1 1" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES &&
run cornell "cornell_box 1024 1024 64 50 26" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES &&
run smoke "cornell_smoke 1024 1024 64 50 26 1 1" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES &&
run rtiow "random_scene 1200 800 50 50 26" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
for n in final cornell smoke rtiow; do python3 tools/pmc_summary.py $OUT/$n.csv $OUT/$n > /dev/null; echo "== $n"; cat $OUT/$n.csv; done
