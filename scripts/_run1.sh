cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --only c5,one_page --no-cpu-baseline > gpurun_out/b_x.json 2> gpurun_out/b_x.err < /dev/null
grep bench gpurun_out/b_x.err | tail -8
