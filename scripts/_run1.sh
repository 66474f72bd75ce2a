cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export SB_ZSTD_BLOCKS=1
timeout 600 python scripts/prof_zstd_ref.py 64 ab 2>&1 | grep -v amdgpu | head -12
timeout 900 python bench.py --only c5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['configs']['c5']
print('c5 enc', d['encode']['GBps'], d['encode']['ms'], 'dec', d['decode']['GBps'], d['decode']['ms'], d['decode']['kernels_ms'], d.get('single_array'))"
