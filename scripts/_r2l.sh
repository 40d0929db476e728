cd $GRAFT_REPO_ROOT
for k in 1 32; do echo "== K=$k"; ./tools/sort_bench_lb$k 1000000 27 2; ./tools/sort_bench_lb$k 4055040 27 2; done > gpurun_out/r2l_lb.log 2>&1
echo "== K=8" >> gpurun_out/r2l_lb.log; ./tools/sort_bench_probe 4055040 27 2 >> gpurun_out/r2l_lb.log 2>&1
