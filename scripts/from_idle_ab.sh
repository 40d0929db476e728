#!/bin/bash
# dev, on the GPU box: bench.py's from-idle leg (0.5 s of sleep, then W + K steps) with the split stage on and off
cd $GRAFT_REPO_ROOT
for SPLIT in 1 0 1 0; do
  echo "== split=$SPLIT"
  GGR_SPLIT_COLOUR=$SPLIT python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1 | grep -E "from idle|timed" | cut -c1-300
done
