#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + two separate PMC passes (FETCH_SIZE / WRITE_SIZE
# cannot share a pass on gfx950: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2) of the bench command.
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $*"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
ls $OUT
