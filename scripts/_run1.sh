cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py > gpurun_out/bench_r04_b.json 2> gpurun_out/bench_r04_b.err < /dev/null
echo rc=$?
tail -30 gpurun_out/bench_r04_b.err
