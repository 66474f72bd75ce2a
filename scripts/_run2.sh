cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tests/probes/fuzz_zb.py 2>&1 | grep -v amdgpu.ids | tail -2
