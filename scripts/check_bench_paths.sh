GVC_BENCH_SAME_DEVICE=1 GVC_BENCH_BACKEND=gloo GVC_PERSIST=0 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-harness 2>/dev/null | grep "^{" > gpurun_out/r04_bench_dry2.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_dry2.json").read())
print("dry run:", d["n_gpus"], d["rccl_ranks"], d["collective_backend"], round(d["value"],2), round(d["offline_utts_per_s"],1), d["config4"]["n_gpus"])
PY
python bench.py --streams 8 --weights bf16_kv --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r04_bench_streams8_bf16_kv.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_streams8_bf16_kv.json").read())
print("streams 8 bf16_kv:", round(d["value"],2), d["config"]["workload"][:90])
PY
python bench.py --weights bf16_kv --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r04_bench_bf16_kv.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_bf16_kv.json").read())
print("1 stream bf16_kv:", round(d["value"],2), round(d["roofline"]["avg_us"],1))
PY
