cd $GRAFT_REPO_ROOT
{
for n in 1 1000 10000 200000 1000000 1146880 4915200; do for m in 2 3; do ./tools/sort_bench $n 27 $m; done; done
./tools/sort_bench 1000000 30 1; ./tools/sort_bench 1000000 12 1; ./tools/sort_bench 3000000 29 1
for k in 1 4 16; do ./tools/sort_bench_lb$k 1000000 27 2; ./tools/sort_bench_lb$k 4915200 27 2; done
} > gpurun_out/r2c_sort.log 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size_parity.py tests/test_gpu_full_size_properties.py tests/test_gpu_sync_free.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2c_pytest.log
python scripts/grad_outliers.py C3 > gpurun_out/r2c_outliers.log 2>&1
python bench.py --no-cpu-baseline --no-secondary --no-callsite > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err
tail -3 gpurun_out/r2c_pytest.log
