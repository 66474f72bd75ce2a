python -m pytest tests/test_gpu_zstd.py tests/test_gpu_nested.py -m gpu -x -q 2>&1 | tail -5
python bench.py --only c5 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err; tail -3 gpurun_out/c5.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/c5.json'))['configs']['c5']
print(json.dumps({k:d[k] for k in ('arrow_MB','page_MB','encode','decode','single_array')}, indent=1))
PY
