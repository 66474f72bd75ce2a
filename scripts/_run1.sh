cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_big_pages.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --only one_page --no-cpu-baseline > gpurun_out/b_x.json 2> gpurun_out/b_x.err < /dev/null
grep bench gpurun_out/b_x.err | head -2
