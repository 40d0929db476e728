#!/bin/bash
# Runs on the GPU box: kernel trace + stats of the C3 bench command → gpurun_out/<tag>/c3_kernel_stats.txt
# usage: scripts/quick_trace.sh <tag> [extra bench args]
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-callsite --no-graph "$@" > $R/gpurun_out/$TAG/trace.log 2>&1
cd $R
python scripts/rocprof_summary.py gpurun_out/$TAG/trace_results.db > gpurun_out/$TAG/c3_kernel_stats.txt 2>&1
rm -f gpurun_out/$TAG/*.db
head -16 gpurun_out/$TAG/c3_kernel_stats.txt
grep stages gpurun_out/$TAG/trace.log
