#!/bin/bash
# Runs on the GPU box (via gpurun): SQ counters of the bench command, four per pass (counter passes carry
# --kernel-trace only).  Summarise with scripts/pmc_sq_summary.py <dir> profiles/<tag>_pmc_sq.json
# usage: scripts/pmc_sq.sh <tag> [bench args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp; cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$TAG; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-callsite $*"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O -o p$i -- $CMD > $O/p$i.log 2>&1
done
ls $O
