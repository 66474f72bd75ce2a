cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_big_pages.py -q --durations=40 > gpurun_out/big_pages_test.txt 2>&1
tail -50 gpurun_out/big_pages_test.txt
