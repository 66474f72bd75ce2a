cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_big_pages.py tests/test_gpu_decode.py tests/test_gpu_select.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
timeout 300 python tests/probes/one_page_time.py 12000000 "low-card" 2>&1 | grep -v amdgpu.ids | head -3
timeout 300 python tests/probes/one_page_time.py 12000000 "utf8 zipf" 2>&1 | grep -v amdgpu.ids | head -3
