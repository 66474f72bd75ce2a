cd $GRAFT_REPO_ROOT
export SB_ZSTD_BLOCKS=1
timeout 600 python scripts/prof_zstd_ref.py 8 a 2>&1 | grep -v amdgpu | head -6
timeout 600 python scripts/prof_zstd_ref.py 8 b 2>&1 | grep -v amdgpu | head -6
timeout 600 python scripts/prof_zstd_ref.py 64 ab 2>&1 | grep -v amdgpu | head -9
timeout 900 python bench.py --only c5 --no-cpu-baseline 2>&1 | tail -4 | cut -c1-2500
