cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python tests/probes/fuzz_zb.py 40 2>&1 | tail -1
