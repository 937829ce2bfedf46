mkdir -p gpurun_out
F="--steps 12 --warmup 3 --no-cpu-baseline --no-offline --no-harness --no-extra --no-cold"
for i in 1 2 3; do
  (cd .ab_old && python bench.py $F 2>/dev/null | tail -1) > gpurun_out/ab_old_$i.json
  python bench.py $F 2>/dev/null | tail -1 > gpurun_out/ab_new_$i.json
done
python - <<'PY'
import json,glob
for k in ("old","new"):
    for f in sorted(glob.glob(f"gpurun_out/ab_{k}_*.json")):
        d=json.loads(open(f).read())
        print(k, d["value"], d.get("first_chunk_latency_ms"), d.get("first_chunk_latency_ms_registered_speaker"))
PY
python -m pytest tests/test_gpu_round6.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3
