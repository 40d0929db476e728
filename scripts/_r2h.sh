cd $GRAFT_REPO_ROOT
for sp in 1 2 4; do
  echo "== XCD_SPLIT $sp"
  GGR_EXTRA_HIPCC_FLAGS=-DGGR_XCD_SPLIT=$sp python -c "from ggrt_official_amd import _build; _build.build_library(force=True)" > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-callsite --no-graph --steps 10 > gpurun_out/r2h_bench_$sp.json 2> gpurun_out/r2h_$sp.err
  grep -E "stages" gpurun_out/r2h_$sp.err | cut -c1-330
  python - <<PY
import json
r=json.load(open("gpurun_out/r2h_bench_$sp.json"))
for k,v in r["secondary"].items(): print(k, v["ms_per_step"]["median"], {a:round(b,3) for a,b in v["stages_ms"].items()})
PY
done > gpurun_out/r2h_split.log 2>&1
python -c "from ggrt_official_amd import _build; _build.build_library(force=True)"
