cd $GRAFT_REPO_ROOT
python scripts/callsite_bench.py > gpurun_out/r2j_callsite.log 2>&1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2j_prof -o v -- python -c "
import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); sys.path.insert(0,'$GRAFT_REPO_ROOT/scripts')
import callsite_bench; print(callsite_bench.measure_views(steps=5, warmup=2))" > $GRAFT_REPO_ROOT/gpurun_out/r2j_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/rocprof_summary.py gpurun_out/r2j_prof/v_results.db > gpurun_out/r2j_stats.txt 2>&1
cat gpurun_out/r2j_callsite.log
