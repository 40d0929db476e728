cd $GRAFT_REPO_ROOT
{ for a in "1000000 27 2 1" "4055040 27 2 1" "4055040 27 2 4" "4055040 27 3 4" "8110080 27 2 8" "40000 27 3 5" "4055040 27 2 3"; do ./tools/sort_bench $a; done; } > gpurun_out/r2q_sort.log 2>&1
python -m pytest tests/test_gpu_views_batched.py tests/test_gpu_parity.py tests/test_gpu_sync_free.py tests/test_gpu_full_size_properties.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r2q_pytest.log
python -c "
import sys; sys.path.insert(0,'scripts')
import callsite_bench, json; print(json.dumps(callsite_bench.measure_views()))" > gpurun_out/r2q_callsite.log 2>&1
cat gpurun_out/r2q_sort.log; tail -2 gpurun_out/r2q_pytest.log; tail -1 gpurun_out/r2q_callsite.log
