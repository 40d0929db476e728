#!/bin/bash
# dev, on the GPU box: the split per-Gaussian stage (colour on the side stream; GGR_COLOUR_FORK = where it starts: 0 beside
# the depth sort, 1 behind it, 2 behind the tile counts; GGR_COLOUR_BLOCKS_PER_CU persistent blocks per CU, 0 = unthrottled)
# against the one-kernel form (GGR_SPLIT_COLOUR=0), same library: stage times + timed step of bench.py.
# usage: scripts/ab_split.sh "<fork list>" "<blocks-per-CU list>" <config> [<config> …]
R=$GRAFT_REPO_ROOT
cd $R
FORKS=$1; shift
LIST=$1; shift
run() { python bench.py --config $1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1 | grep -E "stages|timed|Error|error" | cut -c1-330 | grep -v '^{'; }
for CFG in "$@"; do
  echo "== $CFG one kernel"
  GGR_SPLIT_COLOUR=0 run $CFG
  for F in $FORKS; do for B in $LIST; do
    echo "== $CFG split, fork $F, $B blocks per CU"
    GGR_COLOUR_FORK=$F GGR_COLOUR_BLOCKS_PER_CU=$B run $CFG
  done; done
done
