#!/bin/bash
# One SQ counter pass over the profiler workload with an alternative library: tools/pmc_quick.sh <lib|base> [driver args]
set -e
LIB=$1; shift
cp path-tracer_amd/libptgpu.so /tmp/base.so
[ "$LIB" != base ] && cp "$LIB" path-tracer_amd/libptgpu.so
OUT=gpurun_out/pmcq_$(basename $LIB .so)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/p -o p -- python3 tools/prof_driver.py "$@" > $OUT/log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k -o k -- python3 tools/prof_driver.py "$@" > $OUT/klog 2>&1
python3 tools/pmc_summary.py $OUT/summary.csv $OUT/p > /dev/null
python3 tools/rocprof_summary.py $OUT/k $OUT/kstats.csv > /dev/null
cat $OUT/summary.csv; head -5 $OUT/kstats.csv; tail -1 $OUT/log | cut -c1-200
cp /tmp/base.so path-tracer_amd/libptgpu.so
