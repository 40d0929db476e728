#!/bin/bash
# dev: per library variant under gpurun_variants/ (and the tree's build, "base", before and after), the stage times of
# bench.py for a config (default C5p)
CFG=${1:-C5p}
R=$GRAFT_REPO_ROOT
export GGR_SKIP_SOURCE_HASH=1   # variants carry the hash of their own flags (_build.source_hash)
cp $R/ggrt_official_amd/libggr_raster.so /tmp/base.so
run() { python bench.py --config $CFG --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1 | grep -E "^\[bench.*(stages|timed)" | cut -c1-330; }
echo "== base"; run
for d in $R/gpurun_variants/*/; do
  n=$(basename $d)
  cp $d/libggr_raster.so $R/ggrt_official_amd/libggr_raster.so
  echo "== $n"; run
done
cp /tmp/base.so $R/ggrt_official_amd/libggr_raster.so
echo "== base again"; run
