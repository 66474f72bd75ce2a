cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_zstd_blocks.py tests/test_gpu_zstd.py -x -q 2>&1 | tail -3
export SB_ZSTD_BLOCKS=1
timeout 600 python scripts/prof_zstd_ref.py 1 a 2>&1 | grep -v amdgpu | tail -3
