cd $GRAFT_REPO_ROOT
for sp in 2 4 1; do
  echo "== SCATTER_PARTS $sp"
  GGR_EXTRA_HIPCC_FLAGS=-DGGR_SCATTER_PARTS=$sp python -c "from ggrt_official_amd import _build; _build.build_library(force=True)" > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-callsite --no-graph --steps 10 2>&1 >/dev/null | grep -E "stages|secondary" | cut -c1-330
done > gpurun_out/r2p_parts.log 2>&1
python -c "from ggrt_official_amd import _build; _build.build_library(force=True)"
