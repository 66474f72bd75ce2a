cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_nested.py tests/test_gpu_configs.py tests/test_gpu_io.py tests/test_gpu_file.py tests/test_gpu_dist.py -x -q 2>&1 | tail -8
timeout 900 python bench.py --only c5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['configs']['c5']
print('c5 enc', d['encode']['GBps'], d['encode']['ms'], d['encode']['kernels_ms'])
print('c5 dec', d['decode']['GBps'], d['decode']['ms'], d['decode']['kernels_ms'], d.get('single_array'))"
