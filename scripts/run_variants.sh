#!/bin/bash
# dev: on the GPU box, time every library variant under gpurun_variants/<name>/ with the kernel trace of bench.py
# usage: scripts/run_variants.sh [kernel-name-pattern]
PAT=${1:-bin_}
R=$GRAFT_REPO_ROOT
export GGR_SKIP_SOURCE_HASH=1   # variants carry the hash of their own flags (_build.source_hash)
cp $R/ggrt_official_amd/libggr_raster.so /tmp/base.so
for d in $R/gpurun_variants/*/; do
  n=$(basename $d)
  cp $d/libggr_raster.so $R/ggrt_official_amd/libggr_raster.so
  scripts/quick_trace.sh var_$n > /dev/null 2>&1
  echo "== $n"; grep -E "$PAT" $R/gpurun_out/var_$n/c3_kernel_stats.txt | cut -c1-110
done
cp /tmp/base.so $R/ggrt_official_amd/libggr_raster.so
