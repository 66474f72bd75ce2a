cd $GRAFT_REPO_ROOT
timeout 2400 bash scripts/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py > gpurun_out/bench_r04_e.json 2> gpurun_out/bench_r04_e.err < /dev/null
echo rc=$?
grep bench gpurun_out/bench_r04_e.err
