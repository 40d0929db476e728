#!/bin/bash
# dev: builds one library variant per argument (extra hipcc flags, e.g. "-DSURV_GROUP=8") into
# gpurun_variants/<name>/libggr_raster.so (git-ignored; travels to the GPU box with gpurun), then restores the normal build.
# On the box: scripts/run_variants.sh [kernel pattern] / run_variants_views.sh / run_variants_cfg.sh time every variant.
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  name=$(echo $v | tr -d ' -' | tr '=' '_')
  GGR_EXTRA_HIPCC_FLAGS="$v" python -c "
import ggrt_official_amd._build as b; b.build_library(force=True)"
  mkdir -p gpurun_variants/$name; cp ggrt_official_amd/libggr_raster.so gpurun_variants/$name/
  echo built $name
done
python -c "
import ggrt_official_amd._build as b; b.build_library(force=True)"
