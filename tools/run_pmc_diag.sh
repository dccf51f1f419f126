#!/bin/bash
# Extra diagnostic PMC passes (instruction cache, memory-pipe pressure, LDS, f32 mix). Same conventions as
# run_pmc.sh: tools/run_pmc_diag.sh <tag> [driver args]
set -e
TAG=$1; shift
OUT=gpurun_out/pmcd_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="$*"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/prof_driver.py $ARGS > $OUT/$name.log 2>&1; }
run d1 SQ_WAVE_CYCLES SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS &&
run d2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQ_BUSY_CU_CYCLES &&
run d3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_LOAD &&
run t1 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum &&
run t2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TCP_TCC_WRITE_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
python3 tools/pmc_summary.py $OUT/summary.csv $OUT/d1 $OUT/d2 $OUT/d3 $OUT/t1 $OUT/t2 > /dev/null
cat $OUT/summary.csv
