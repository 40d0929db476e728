#!/bin/bash
# dev, on the GPU box: large / multi-view shapes under the colour kernel's throttle settings
cd $GRAFT_REPO_ROOT
run() { python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1 | grep -E "^\[bench.*(stages|timed)" | cut -c1-330; }
for CFG in C6p; do
  echo "== $CFG one kernel"; GGR_SPLIT_COLOUR=0 run $CFG
  echo "== $CFG adaptive"; run $CFG
  for B in 1 2 4; do echo "== $CFG $B per CU"; GGR_COLOUR_BLOCKS_PER_CU=$B run $CFG; done
done
for SPLIT in 0 1; do
  echo "== views4/sets4/eval split=$SPLIT"
  GGR_SPLIT_COLOUR=$SPLIT python -c "
import sys; sys.path.insert(0, 'scripts')
import callsite_bench as c
print('views4', c.measure_views(steps=20, warmup=5, only='batched_ms'))
print('sets4', c.measure_sets(steps=20, warmup=5))
print('eval', c.measure_eval_sets(steps=20, warmup=5))
" 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-400
done
