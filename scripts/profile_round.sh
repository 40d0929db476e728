#!/bin/bash
# Runs on the GPU box (via gpurun): everything the round's profiles/ files come from, in one call.
#   kernel trace + stats of the C3 bench command, the two HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate: TCC
#   has 4 counter slots), four SQ counter passes, and the kernel stats of the 4-view batched call site at GGRt's shape.
# Counter passes carry --kernel-trace only (never sys/runtime trace together with --pmc).
# usage: scripts/profile_round.sh <tag> [config, default C3]   → gpurun_out/prof_<tag>/…  (summaries: *.txt / *.json, copy to profiles/)
set -u
TAG=$1
CFG=${2:-C3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-callsite --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- $BENCH > $OUT/write.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o p$i -- $BENCH --steps 3 --warmup 2 > $OUT/p$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o views -- python -c "
import sys; sys.path.insert(0,'$R'); sys.path.insert(0,'$R/scripts')
import callsite_bench, torch
from ggrt_official_amd import splatting
print(callsite_bench.measure_views(steps=10, warmup=3))" > $OUT/views.log 2>&1
cd $R
# which kernels these profiles were taken on: bench.py compares the hash with csrc/ and flags a stale profile
python -c "
import json, sys; sys.path.insert(0, '$R')
from ggrt_official_amd import _build
json.dump({'source_hash': _build.source_hash(), 'git_sha': '${GGR_GIT_SHA:-unknown}', 'tag': '$TAG', 'config': '$CFG'}, open('$OUT/meta.json', 'w'))"
python scripts/rocprof_summary.py $OUT/trace_results.db > $OUT/c3_kernel_stats.txt 2>&1
python scripts/rocprof_summary.py $OUT/views_results.db > $OUT/views4_c5p_kernel_stats.txt 2>&1
python scripts/pmc_summary.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python scripts/pmc_sq_summary.py $OUT $OUT/pmc_sq.json > $OUT/pmc_sq.txt 2>&1
rm -f $OUT/*.db $OUT/*_kernel_trace.csv $OUT/*agent_info.csv
ls $OUT
