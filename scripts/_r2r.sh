cd $GRAFT_REPO_ROOT
for sl in 16 8 4; do
  echo "== CKPT_MID $sl"
  GGR_EXTRA_HIPCC_FLAGS=-DGGR_CKPT_MID=$sl python -c "from ggrt_official_amd import _build; _build.build_library(force=True)" > /dev/null 2>&1
  python -c "
import sys; sys.path.insert(0,'scripts')
import callsite_bench, json; print(json.dumps(callsite_bench.measure_views()))" 2>&1 | tail -1
done > gpurun_out/r2r_ckpt.log 2>&1
python -c "from ggrt_official_amd import _build; _build.build_library(force=True)"
