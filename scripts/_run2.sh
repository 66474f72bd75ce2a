cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_nested.py tests/test_gpu_io.py -x -q 2>&1 | tail -2
timeout 300 python scripts/prof_c5_host.py 64 2>&1 | grep -E "per run|_levels|synchronize|nested.py:96"
