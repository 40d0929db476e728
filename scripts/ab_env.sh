#!/bin/bash
# dev: on the GPU box, bench.py (C3, 100 steps) under each environment setting given as an argument ("A=1 B=2"), one line each:
# the per-stage HIP-event times and the step.  usage: scripts/ab_env.sh "GGR_COLOUR_FORK=0" "GGR_COLOUR_FORK=2 GGR_COLOUR_BLOCKS_PER_CU=2" …
R=$GRAFT_REPO_ROOT
CFG=${AB_CONFIG:-C3}
for e in "$@"; do
  out=$(env $e python $R/bench.py --config $CFG --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1)
  st=$(echo "$out" | grep "stages:" | sed 's/.*stages: //; s/fwd_//g; s/_ms=/=/g')
  ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1)
  fw=$(echo "$out" | grep -o '"t_fwd_ms_events": [0-9.]*' | head -1)
  echo "== $e | $ms | $fw | $st"
done
