cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_views_batched.py tests/test_gpu_camera_and_depth_grads.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r2k_views.log
python -c "
import sys; sys.path.insert(0,'scripts')
import callsite_bench, json; print(json.dumps(callsite_bench.measure_views()))" > gpurun_out/r2k_callsite.log 2>&1
tail -3 gpurun_out/r2k_views.log; cat gpurun_out/r2k_callsite.log
