cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2t_pytest.log
python bench.py --no-cpu-baseline --no-graph > gpurun_out/r2t_bench.json 2> gpurun_out/r2t_bench.err
python -c "
import sys; sys.path.insert(0,'scripts')
import callsite_bench, json; print(json.dumps(callsite_bench.measure_views()))" > gpurun_out/r2t_callsite.log 2>&1
tail -4 gpurun_out/r2t_pytest.log; grep -E "stages|secondary|call site" gpurun_out/r2t_bench.err | cut -c1-330; tail -1 gpurun_out/r2t_callsite.log
