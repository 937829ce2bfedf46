export TMPDIR=/tmp
python -m pytest tests/test_gpu_shell.py tests/test_gpu_rows_persist.py -x -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline > gpurun_out/bench_e2e.json 2> gpurun_out/bench_e2e.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_e2e.json').read().strip().splitlines()[-1])
print(d['value'], d['first_chunk_latency_ms'], d['offline'])
PY
