cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_configs.py -x -q 2>&1 | tail -6
( time timeout 1500 python bench.py > gpurun_out/bench_r04_a.json 2> gpurun_out/bench_r04_a.err ) 2>&1 | tail -4
tail -c 3000 gpurun_out/bench_r04_a.json
grep -v "^\[bench\]" gpurun_out/bench_r04_a.err | tail -5
