cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_views_batched.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r2i_views.log
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r2i_pytest.log
tail -5 gpurun_out/r2i_views.log; tail -5 gpurun_out/r2i_pytest.log
