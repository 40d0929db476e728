#!/bin/bash
# dev: per library variant under gpurun_variants/, the stage times of bench.py for a config (default C5p) and of C4p forward
CFG=${1:-C5p}
R=$GRAFT_REPO_ROOT
export GGR_SKIP_SOURCE_HASH=1   # variants carry the hash of their own flags (_build.source_hash)
cp $R/ggrt_official_amd/libggr_raster.so /tmp/base.so
for d in $R/gpurun_variants/*/; do
  n=$(basename $d)
  cp $d/libggr_raster.so $R/ggrt_official_amd/libggr_raster.so
  echo "== $n"
  python bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1 | grep -E "stages|timed"
done
cp /tmp/base.so $R/ggrt_official_amd/libggr_raster.so
