#!/bin/bash
# dev: per library variant under gpurun_variants/, the deferred-back-propagation cell of callsite_bench.measure_window (config $1, default C5p)
CFG=${1:-C5p}
R=$GRAFT_REPO_ROOT
export GGR_SKIP_SOURCE_HASH=1   # variants carry the hash of their own flags (_build.source_hash)
cp $R/ggrt_official_amd/libggr_raster.so /tmp/base.so
for d in $R/gpurun_variants/*/; do
  n=$(basename $d)
  cp $d/libggr_raster.so $R/ggrt_official_amd/libggr_raster.so
  echo "== $n"
  python -c "
import sys; sys.path.insert(0,'$R'); sys.path.insert(0,'$R/scripts')
import callsite_bench, json
r = callsite_bench.measure_window(config='$CFG')
for k, v in r.items(): print(' ', k, v)"
done
cp /tmp/base.so $R/ggrt_official_amd/libggr_raster.so
