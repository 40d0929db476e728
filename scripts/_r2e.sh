cd $GRAFT_REPO_ROOT
{
for n in 1 1000 4097 200000 1000000 1146880 4915200; do for m in 2 3; do ./tools/sort_bench $n 27 $m; done; done
./tools/sort_bench 1000000 30 1; ./tools/sort_bench 1000000 12 1
./tools/sort_bench_probe 1000000 27 2
} > gpurun_out/r2e_sort.log 2>&1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2e_prof -o s -- $GRAFT_REPO_ROOT/tools/sort_bench 1000000 27 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python scripts/rocprof_summary.py gpurun_out/r2e_prof/s_results.db > gpurun_out/r2e_stats.txt 2>&1
