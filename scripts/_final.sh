cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/final_tests.txt
timeout 600 python bench.py > gpurun_out/bench_r06_final.json 2> gpurun_out/bench_r06_final.err
timeout 2000 bash scripts/profile_round.sh r06 > gpurun_out/prof_r06.log 2>&1
timeout 300 python bench.py --no-configs > gpurun_out/bench_r06_final2.json 2> gpurun_out/bench_r06_final2.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.txt 2>&1
cat gpurun_out/final_tests.txt; tail -2 gpurun_out/final_smoke.txt; tail -3 gpurun_out/prof_r06.log
