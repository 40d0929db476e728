cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_views_batched.py tests/test_gpu_camera_and_depth_grads.py tests/test_callsite_fused.py tests/test_adapter_fusion.py tests/test_gpu_depth_segments.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r2n_pytest.log
python bench.py --no-cpu-baseline --no-callsite --no-graph > gpurun_out/r2n_bench.json 2> gpurun_out/r2n_bench.err
tail -2 gpurun_out/r2n_pytest.log; grep -E "stages|secondary" gpurun_out/r2n_bench.err | cut -c1-330
