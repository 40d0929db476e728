#!/bin/bash
# dev: per library variant under gpurun_variants/, run a pytest selection ($1) and a bench config ($2, default C6p)
SEL=$1; CFG=${2:-C6p}
R=$GRAFT_REPO_ROOT
export GGR_SKIP_SOURCE_HASH=1   # variants carry the hash of their own flags (_build.source_hash)
cp $R/ggrt_official_amd/libggr_raster.so /tmp/base.so
for d in $R/gpurun_variants/*/; do
  n=$(basename $d)
  cp $d/libggr_raster.so $R/ggrt_official_amd/libggr_raster.so
  echo "== $n"
  python -m pytest $SEL -x -q 2>&1 | tail -3
  python bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-callsite --no-graph 2>&1 | grep -E "stages|timed"
done
cp /tmp/base.so $R/ggrt_official_amd/libggr_raster.so
