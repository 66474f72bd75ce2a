cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_zstd.py -x -q 2>&1 | tail -2
timeout 200 python bench.py --only continuity --no-cpu-baseline 2>&1 | tail -1 | timeout 20 python -c "
import json,sys
d=json.loads(sys.stdin.read())['configs']['continuity']
for k in ('bool','i64','utf8'):
    print(k, 'dec', d[k]['decode']['GBps'], d[k]['decode']['ms'], d[k]['decode']['kernels_ms'])"
