cd $GRAFT_REPO_ROOT
timeout 600 python tests/probes/fuzz_zstd_frames.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python bench.py --only one_page --no-cpu-baseline > gpurun_out/b_one_page.json 2> gpurun_out/b_one_page.err
tail -12 gpurun_out/b_one_page.err
