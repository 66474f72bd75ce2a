cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --only one_page,c1 --no-cpu-baseline > gpurun_out/b_x.json 2> gpurun_out/b_x.err < /dev/null
grep bench gpurun_out/b_x.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
