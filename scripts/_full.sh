export TMPDIR=/tmp
python bench.py --no-cpu-baseline > gpurun_out/bench_e2e.json 2> gpurun_out/bench_e2e.err
python bench.py --streams 8 --no-cpu-baseline --no-offline --no-harness > gpurun_out/bench_streams8.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_e2e.json').read().strip().splitlines()[-1])
print(d['value'], d['first_chunk_latency_ms'], d['offline_utts_per_s'], d['offline']['offline_utts_per_s_fully_batched_1gpu'], d['roofline']['decode_step_us'])
d=json.loads(open('gpurun_out/bench_streams8.json').read().strip().splitlines()[-1])
print(d['value'], d['first_chunk_latency_ms'])
PY
python scripts/time_prefill_cached.py 2>&1 | tail -2
