#!/bin/bash
# dev, on the GPU box: the per-quadrant forward blend (GGR_BLEND_FWD_QUADRANTS=1) against the per-tile one (0): stage times
cd $GRAFT_REPO_ROOT
run() { python bench.py --config $1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1 | grep -E "^\[bench.*(stages|timed)" | cut -c1-330; }
for CFG in "$@"; do for Q in 0 1 0 1; do
  echo "== $CFG quadrants=$Q"; GGR_BLEND_FWD_QUADRANTS=$Q run $CFG
done; done
