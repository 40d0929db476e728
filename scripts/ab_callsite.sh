#!/bin/bash
# dev, on the GPU box: the call-site legs (one view fused, 4 views, 4 sets, eval 4 sets) under SH cap 3 / 4 and with the split
# per-Gaussian stage on / off
cd $GRAFT_REPO_ROOT
for CAP in 3 4; do for SPLIT in 0 1; do
  echo "== cap $CAP split $SPLIT"
  GGR_SH_MAX_DEGREE=$CAP GGR_SPLIT_COLOUR=$SPLIT python -c "
import sys; sys.path.insert(0, 'scripts')
import callsite_bench as c
m = c.measure(steps=20, warmup=5); print('one view', m)
print('views4', c.measure_views(steps=20, warmup=5))
print('sets4', c.measure_sets(steps=20, warmup=5))
print('eval', c.measure_eval_sets(steps=20, warmup=5))
" 2>&1 | grep -v Warning
done; done
