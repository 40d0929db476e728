cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size_properties.py tests/test_gpu_sync_free.py tests/test_gpu_camera_setup.py tests/test_gpu_depth_segments.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r2f_pytest.log
python bench.py --no-cpu-baseline --no-callsite > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
tail -3 gpurun_out/r2f_pytest.log; grep -E "stages|secondary" gpurun_out/r2f_bench.err | cut -c1-400
