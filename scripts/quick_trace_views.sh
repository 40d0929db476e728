#!/bin/bash
# GPU box: kernel stats of the 4-view batched call site at GGRt's shape → gpurun_out/<tag>/views4_kernel_stats.txt
TAG=$1; WHAT=${2:-measure_views}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o views -- python -c "
import sys; sys.path.insert(0,'$R'); sys.path.insert(0,'$R/scripts')
import callsite_bench
print(callsite_bench.$WHAT(steps=10, warmup=3))" > $R/gpurun_out/$TAG/views.log 2>&1
cd $R
python scripts/rocprof_summary.py gpurun_out/$TAG/views_results.db > gpurun_out/$TAG/${WHAT}_kernel_stats.txt 2>&1
rm -f gpurun_out/$TAG/*.db
head -24 gpurun_out/$TAG/${WHAT}_kernel_stats.txt | cut -c1-125
tail -2 gpurun_out/$TAG/views.log
