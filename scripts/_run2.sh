cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tests/probes/one_page_time.py 12000000 "bool" 2>&1 | grep -v amdgpu.ids | head -3
