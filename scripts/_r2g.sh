cd $GRAFT_REPO_ROOT
for nb in 8 16 24 32 64; do
  echo "== bands $nb" 
  GGR_SCATTER_BANDS=$nb python bench.py --no-cpu-baseline --no-callsite --no-graph --steps 10 2>&1 >/dev/null | grep -E "stages|secondary C3_lower|secondary C5p" | cut -c1-330
done > gpurun_out/r2g_bands.log 2>&1
