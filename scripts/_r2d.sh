cd $GRAFT_REPO_ROOT
./tools/sort_bench_probe 1000000 27 2 > gpurun_out/r2d_probe.log 2>&1
./tools/sort_bench_probe 4915200 27 2 >> gpurun_out/r2d_probe.log 2>&1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2d_prof -o s -- $GRAFT_REPO_ROOT/tools/sort_bench 1000000 27 2 > $GRAFT_REPO_ROOT/gpurun_out/r2d_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/rocprof_summary.py gpurun_out/r2d_prof > gpurun_out/r2d_stats.txt 2>&1 || ls -R gpurun_out/r2d_prof | head
