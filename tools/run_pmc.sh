#!/bin/bash
# Collects PMC counters for the profiler workload, one rocprofv3 run per counter set (the hardware holds at most
# 8 SQ / 4 TCC counters per pass). Run on the GPU box from the repository root: tools/run_pmc.sh <tag> [driver args]
set -e
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/prof_driver.py $ARGS > $OUT/$name.log 2>&1; }
ARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 &&
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA &&
run tcc1 FETCH_SIZE TCC_HIT_sum &&
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 tools/pmc_summary.py $OUT/summary.csv $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/tcc1 $OUT/tcc2 $OUT/grbm > /dev/null
cat $OUT/summary.csv
