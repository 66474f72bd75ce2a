cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --only c5,one_page --no-cpu-baseline > gpurun_out/b_x.json 2> gpurun_out/b_x.err < /dev/null
grep bench gpurun_out/b_x.err | tail -8
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstd_blocks.py tests/test_gpu_nested.py tests/test_gpu_configs.py tests/test_gpu_io.py tests/test_gpu_file.py tests/test_gpu_big_pages.py -x -q 2>&1 | tail -3
