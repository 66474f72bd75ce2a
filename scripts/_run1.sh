cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_zstd_blocks.py tests/test_gpu_zstd.py -x -q 2>&1 | tail -5
timeout 900 python tests/probes/fuzz_zb.py 60 2>&1 | tail -2
export SB_ZSTD_BLOCKS=1
timeout 600 python scripts/prof_zstd_ref.py 8 b 2>&1 | grep -v amdgpu | head -8
timeout 600 python scripts/prof_zstd_ref.py 64 ab 2>&1 | grep -v amdgpu | head -10
