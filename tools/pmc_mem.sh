#!/bin/bash
# Vector-memory path counters (address unit, L1, data return) of the profiler workload, one rocprofv3 run per set:
# tools/pmc_mem.sh <tag> [driver args]
set -e
TAG=$1; shift
OUT=gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/prof_driver.py $ARGS > $OUT/$name.log 2>&1; }
ARGS="$*"
run m1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD &&
run m2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum &&
run m3 TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_HIT_sum SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS
python3 tools/pmc_summary.py $OUT/summary.csv $OUT/m1 $OUT/m2 $OUT/m3 > /dev/null
cat $OUT/summary.csv
